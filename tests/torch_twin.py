"""Independent second opinion for the C oracle: the same published algorithms
(kaldi fbank; icefall Zipformer2 / stateless decoder / joiner inference graphs)
written with numpy / torch-CPU library ops in icefall's own (T, B, D) layout.

It shares no code with oracle/k2_oracle.c: conv2d/conv1d/softmax/matmul come
from torch, the FFT from numpy (float64).  tests/ compares the two so that a
slip in the hand-written C loops (an index, a stride, a missed bias) shows up
on the CPU, before the HIP kernels are compared against that C.
Not the reference (which cannot run here) and never shipped in the product.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _ints(meta, key):
    return [int(x) for x in meta[key].split(",") if x]


# ----------------------------------------------------------------------------- fbank
def fbank_np(samples: np.ndarray, meta: dict) -> np.ndarray:
    sr = int(meta["sample_rate"])
    N = sr * int(meta["frame_length_ms"]) // 1000
    S = sr * int(meta["frame_shift_ms"]) // 1000
    P = 1 << (N - 1).bit_length()
    nb = int(meta["feature_dim"])
    x = samples.astype(np.float64) * float(meta["input_scale"])
    if x.size < N:
        return np.zeros((0, nb), np.float32)
    nf = 1 + (x.size - N) // S
    idx = np.arange(N)[None, :] + S * np.arange(nf)[:, None]
    fr = x[idx]
    if int(meta["remove_dc_offset"]):
        fr = fr - fr.mean(axis=1, keepdims=True)
    c = float(meta["preemph_coeff"])
    if c != 0.0:
        prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
        fr = fr - c * prev
    a = 2.0 * math.pi / (N - 1)
    i = np.arange(N)
    wt = meta["window_type"]
    if wt == "hamming":
        w = 0.54 - 0.46 * np.cos(a * i)
    elif wt == "hanning":
        w = 0.5 - 0.5 * np.cos(a * i)
    elif wt == "povey":
        w = (0.5 - 0.5 * np.cos(a * i)) ** 0.85
    else:
        w = np.ones(N)
    fr = fr * w
    spec = np.fft.rfft(fr, n=P, axis=1)
    pw = (spec.real**2 + spec.imag**2)[:, : P // 2]

    def mel(f):
        return 1127.0 * np.log(1.0 + f / 700.0)

    lo, hi = float(meta["low_freq"]), float(meta["high_freq"])
    if hi <= 0:
        hi += sr / 2
    ml, mh = mel(lo), mel(hi)
    d = (ml_d := (mh - ml) / (nb + 1))
    fm = mel(np.arange(P // 2) * (sr / P))
    W = np.zeros((nb, P // 2))
    for b in range(nb):
        l, c_, r = ml + b * d, ml + (b + 1) * d, ml + (b + 2) * d
        up = (fm - l) / (c_ - l)
        dn = (r - fm) / (r - c_)
        sel = (fm > l) & (fm < r)
        W[b, sel] = np.where(fm[sel] <= c_, up[sel], dn[sel])
    e = pw @ W.T
    e = np.maximum(e, np.finfo(np.float32).eps)
    return np.log(e).astype(np.float32)


# ----------------------------------------------------------------------------- activations
def swoosh_l(x):
    return torch.logaddexp(torch.zeros_like(x), x - 4.0) - 0.08 * x - 0.035


def swoosh_r(x):
    return torch.logaddexp(torch.zeros_like(x), x - 1.0) - 0.08 * x - 0.313261687


def bias_norm(x, bias, log_scale):
    scales = (torch.mean((x - bias) ** 2, dim=-1, keepdim=True) ** -0.5) * log_scale.exp()
    return x * scales


class Twin:
    def __init__(self, meta: dict, tensors: dict):
        self.meta = meta
        self.w = {k: torch.from_numpy(np.array(v)) for k, v in tensors.items()}
        self.dims = _ints(meta, "encoder_dims")
        self.layers = _ints(meta, "num_encoder_layers")
        self.heads = _ints(meta, "num_heads")
        self.dss = _ints(meta, "downsampling_factors")
        self.qhd = _ints(meta, "query_head_dims")
        self.phd = _ints(meta, "pos_head_dims")
        self.vhd = _ints(meta, "value_head_dims")
        self.kern = _ints(meta, "cnn_module_kernels")
        self.pos_dim = int(meta["pos_dim"])
        self.ctx = int(meta["context_size"])

    def lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    # ---- Conv2dSubsampling
    def encoder_embed(self, x):  # x: (N, T, F)
        w = self.w
        e = "encoder_embed."
        x = x.unsqueeze(1)
        x = swoosh_r(F.conv2d(x, w[e + "conv.0.weight"], w[e + "conv.0.bias"], padding=(0, 1)))
        x = swoosh_r(F.conv2d(x, w[e + "conv.4.weight"], w[e + "conv.4.bias"], stride=2))
        x = swoosh_r(F.conv2d(x, w[e + "conv.7.weight"], w[e + "conv.7.bias"], stride=(1, 2)))
        byp = x
        y = F.conv2d(x, w[e + "convnext.depthwise_conv.weight"], w[e + "convnext.depthwise_conv.bias"], padding=(3, 3), groups=128)
        y = F.conv2d(y, w[e + "convnext.pointwise_conv1.weight"], w[e + "convnext.pointwise_conv1.bias"])
        y = swoosh_l(y)
        y = F.conv2d(y, w[e + "convnext.pointwise_conv2.weight"], w[e + "convnext.pointwise_conv2.bias"])
        x = byp + y
        b, c, t, f = x.shape
        x = x.transpose(1, 2).reshape(b, t, c * f)
        x = self.lin(x, e + "out")
        return bias_norm(x, w[e + "out_norm.bias"], w[e + "out_norm.log_scale"][0])

    # ---- CompactRelPositionalEncoding
    def pos_emb(self, T):
        D = self.pos_dim
        x = torch.arange(-(T - 1), T).to(torch.float32).unsqueeze(1)
        freqs = 1 + torch.arange(D // 2)
        cl = D**0.5
        xc = cl * x.sign() * ((x.abs() + cl).log() - math.log(cl))
        ls = D / (2.0 * math.pi)
        xa = (xc / ls).atan()
        pe = torch.zeros(x.shape[0], D)
        pe[:, 0::2] = (xa * freqs).cos()
        pe[:, 1::2] = (xa * freqs).sin()
        pe[:, -1] = 1.0
        return pe.unsqueeze(0)  # (1, 2T-1, D)

    def attn_weights(self, si, p, x, pos_emb):
        H, q, ph = self.heads[si], self.qhd[si], self.phd[si]
        x = self.lin(x, p + "self_attn_weights.in_proj")
        T, B, _ = x.shape
        qd = q * H
        qq, kk, pp = x[..., :qd], x[..., qd : 2 * qd], x[..., 2 * qd :]
        qq = qq.reshape(T, B, H, q).permute(2, 1, 0, 3)
        pp = pp.reshape(T, B, H, ph).permute(2, 1, 0, 3)
        kk = kk.reshape(T, B, H, q).permute(2, 1, 3, 0)
        scores = torch.matmul(qq, kk)
        pe = F.linear(pos_emb, self.w[p + "self_attn_weights.linear_pos.weight"])
        pe = pe.reshape(-1, 2 * T - 1, H, ph).permute(2, 0, 3, 1)
        ps = torch.matmul(pp, pe)  # (H, B, T, 2T-1)
        ps = ps.as_strided((H, B, T, T), (ps.stride(0), ps.stride(1), ps.stride(2) - ps.stride(3), ps.stride(3)),
                           storage_offset=ps.stride(3) * (T - 1))
        return (scores + ps).softmax(dim=-1)

    def self_attn(self, si, p, k, x, aw):
        T, B, _ = x.shape
        H = aw.shape[0]
        v = self.lin(x, p + f"self_attn{k}.in_proj").reshape(T, B, H, -1).permute(2, 1, 0, 3)
        v = torch.matmul(aw, v).permute(2, 1, 0, 3).reshape(T, B, -1)
        return self.lin(v, p + f"self_attn{k}.out_proj")

    def ff(self, p, k, x):
        return self.lin(swoosh_l(self.lin(x, p + f"feed_forward{k}.in_proj")), p + f"feed_forward{k}.out_proj")

    def nonlin(self, p, x, aw0):
        x = self.lin(x, p + "nonlin_attention.in_proj")
        T, B, _ = x.shape
        s, xx, y = x.chunk(3, dim=2)
        xx = xx * torch.tanh(s)
        xx = xx.reshape(T, B, 1, -1).permute(2, 1, 0, 3)
        xx = torch.matmul(aw0, xx).permute(2, 1, 0, 3).reshape(T, B, -1)
        return self.lin(xx * y, p + "nonlin_attention.out_proj")

    def conv_module(self, si, p, k, x):
        K = self.kern[si]
        x = self.lin(x, p + f"conv_module{k}.in_proj")
        x, s = x.chunk(2, dim=2)
        x = (x * torch.sigmoid(s)).permute(1, 2, 0)
        x = F.conv1d(x, self.w[p + f"conv_module{k}.depthwise_conv.weight"], self.w[p + f"conv_module{k}.depthwise_conv.bias"],
                     padding=K // 2, groups=x.shape[1])
        x = x.permute(2, 0, 1)
        return self.lin(swoosh_r(x), p + f"conv_module{k}.out_proj")

    def layer(self, si, li, src, pos_emb):
        p = f"encoder.encoders.{si}.layers.{li}."
        w = self.w
        orig = src
        aw = self.attn_weights(si, p, src, pos_emb)
        src = src + self.ff(p, 1, src)
        src = src + self.nonlin(p, src, aw[0:1])
        src = src + self.self_attn(si, p, 1, src, aw)
        src = src + self.conv_module(si, p, 1, src)
        src = src + self.ff(p, 2, src)
        src = orig + (src - orig) * w[p + "bypass_mid.bypass_scale"]
        src = src + self.self_attn(si, p, 2, src, aw)
        src = src + self.conv_module(si, p, 2, src)
        src = src + self.ff(p, 3, src)
        src = bias_norm(src, w[p + "norm.bias"], w[p + "norm.log_scale"][0])
        return orig + (src - orig) * w[p + "bypass.bypass_scale"]

    @staticmethod
    def downsample(src, bias, ds):
        T, B, D = src.shape
        Td = (T + ds - 1) // ds
        pad = Td * ds - T
        src = torch.cat((src, src[T - 1 :].expand(pad, B, D)), dim=0).reshape(Td, ds, B, D)
        wts = bias.softmax(dim=0).unsqueeze(-1).unsqueeze(-1)
        return (src * wts).sum(dim=1)

    def stack(self, si, x):
        ds = self.dss[si]
        if ds == 1:
            pe = self.pos_emb(x.shape[0])
            for li in range(self.layers[si]):
                x = self.layer(si, li, x, pe)
            return x
        st = f"encoder.encoders.{si}."
        orig = x
        x = self.downsample(x, self.w[st + "downsample.bias"], ds)
        pe = self.pos_emb(x.shape[0])
        for li in range(self.layers[si]):
            x = self.layer(si, li, x, pe)
        T, B, D = x.shape
        x = x.unsqueeze(1).expand(T, ds, B, D).reshape(T * ds, B, D)[: orig.shape[0]]
        return orig + (x - orig) * self.w[st + "out_combiner.bypass_scale"]

    def encoder(self, feats, tap=None):  # feats (N, T, F) -> (N, T', J)
        x = self.encoder_embed(feats)
        if tap == 0:
            return x
        x = x.permute(1, 0, 2)
        outs = []
        for si, D in enumerate(self.dims):
            c = x.shape[-1]
            x = x[..., :D] if D <= c else F.pad(x, (0, D - c))
            x = self.stack(si, x)
            outs.append(x)
            if tap == 1 + si:
                return x.permute(1, 0, 2)
        pieces = [outs[-1]]
        cur = self.dims[-1]
        for i in range(len(self.dims) - 2, -1, -1):
            d = self.dims[i]
            if d > cur:
                pieces.append(outs[i][..., cur:d])
                cur = d
        x = torch.cat(pieces, dim=-1)
        if tap == 100:
            return x.permute(1, 0, 2)
        x = self.downsample(x, self.w["encoder.downsample_output.bias"], 2)
        return self.lin(x.permute(1, 0, 2), "joiner.encoder_proj")

    def decoder(self, y):  # (N, ctx) int64
        emb = self.w["decoder.embedding.weight"]
        e = emb[y.clamp(min=0)] * (y >= 0).unsqueeze(-1)
        e = e.permute(0, 2, 1)
        e = F.conv1d(e, self.w["decoder.conv.weight"], groups=emb.shape[1] // 4)
        e = F.relu(e.permute(0, 2, 1)).squeeze(1)
        return self.lin(e, "joiner.decoder_proj")

    def joiner(self, enc, dec):
        return self.lin(torch.tanh(enc + dec), "joiner.output_linear")
