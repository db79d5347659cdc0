"""Independent torch restatement of the icefall Zipformer2 *streaming* inference graph
(zipformer.py *.streaming_forward + export-onnx-streaming.py OnnxEncoder.forward), in icefall's
own (T, B, D) layout with library conv / matmul / softmax.  Shares no code with
oracle/k2_oracle_online.c; used by the CPU tests to pin the C oracle from outside."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from torch_twin import Twin, bias_norm, swoosh_l, swoosh_r


class OnlineTwin(Twin):
    def __init__(self, meta, tensors):
        super().__init__(meta, tensors)
        self.left = [int(x) for x in meta["left_context_len"].split(",")]
        self.T = int(meta["T"])

    def init_states(self, B=1):
        st = []
        for si, D in enumerate(self.dims):
            H, L, K = self.heads[si], self.left[si], self.kern[si]
            for _ in range(self.layers[si]):
                st += [torch.zeros(L, B, self.qhd[si] * H), torch.zeros(1, B, L, 3 * D // 4), torch.zeros(L, B, self.vhd[si] * H),
                       torch.zeros(L, B, self.vhd[si] * H), torch.zeros(B, D, K // 2), torch.zeros(B, D, K // 2)]
        st += [torch.zeros(B, 128, 3, 19), torch.zeros(B, dtype=torch.int64)]
        return st

    def embed_stream(self, x, cache):
        w, e = self.w, "encoder_embed."
        x = x.unsqueeze(1)
        x = swoosh_r(F.conv2d(x, w[e + "conv.0.weight"], w[e + "conv.0.bias"], padding=(0, 1)))
        x = swoosh_r(F.conv2d(x, w[e + "conv.4.weight"], w[e + "conv.4.bias"], stride=2))
        x = swoosh_r(F.conv2d(x, w[e + "conv.7.weight"], w[e + "conv.7.bias"], stride=(1, 2)))
        T = x.size(2) - 3
        bypass = x[:, :, :T, :]
        x = torch.cat([cache, x], dim=2)
        cache = x[:, :, T : 3 + T, :]
        y = F.conv2d(x, w[e + "convnext.depthwise_conv.weight"], w[e + "convnext.depthwise_conv.bias"], padding=(0, 3), groups=128)
        y = F.conv2d(y, w[e + "convnext.pointwise_conv1.weight"], w[e + "convnext.pointwise_conv1.bias"])
        y = F.conv2d(swoosh_l(y), w[e + "convnext.pointwise_conv2.weight"], w[e + "convnext.pointwise_conv2.bias"])
        x = bypass + y
        b, c, t, f = x.shape
        x = self.lin(x.transpose(1, 2).reshape(b, t, c * f), e + "out")
        return bias_norm(x, w[e + "out_norm.bias"], w[e + "out_norm.log_scale"][0]), cache

    def pos_emb_stream(self, Tc, L):
        import math
        D = self.pos_dim
        T = Tc + L
        x = torch.arange(-(T - 1), T).to(torch.float32).unsqueeze(1)
        freqs = 1 + torch.arange(D // 2)
        cl = D**0.5
        xc = cl * x.sign() * ((x.abs() + cl).log() - math.log(cl))
        xa = (xc / (D / (2.0 * math.pi))).atan()
        pe = torch.zeros(x.shape[0], D)
        pe[:, 0::2] = (xa * freqs).cos()
        pe[:, 1::2] = (xa * freqs).sin()
        pe[:, -1] = 1.0
        c = pe.size(0) // 2
        return pe[c - T + 1 : c + Tc].unsqueeze(0)

    def conv_stream(self, si, p, k, x, cache):
        K = self.kern[si]
        dc = p + f"conv_module{k}.depthwise_conv."
        x = self.lin(x, p + f"conv_module{k}.in_proj")
        x, s = x.chunk(2, dim=2)
        x = (x * torch.sigmoid(s)).permute(1, 2, 0)
        seq = x.shape[2]
        x = torch.cat([cache, x], dim=2)
        cache = x[..., -(K // 2):]
        xc = F.conv1d(x, self.w[dc + "causal_conv.weight"], self.w[dc + "causal_conv.bias"], groups=x.shape[1])
        xw = F.conv1d(x[..., K // 2:], self.w[dc + "chunkwise_conv.weight"], self.w[dc + "chunkwise_conv.bias"], padding=K // 2, groups=x.shape[1])
        sc = self.w[dc + "chunkwise_conv_scale"]
        le, re = sc[0], sc[1]
        if seq < K:
            le, re = le[:, :seq], re[:, -seq:]
        else:
            pad = torch.zeros(le.shape[0], seq - K)
            le, re = torch.cat((le, pad), -1), torch.cat((pad, re), -1)
        x = xw * (1.0 + (le + re)) + xc
        x = x.permute(2, 0, 1)
        return self.lin(swoosh_r(x), p + f"conv_module{k}.out_proj"), cache

    def layer_stream(self, si, li, src, pos_emb, st, L, mask):
        p = f"encoder.encoders.{si}.layers.{li}."
        w = self.w
        ck, cn, cv1, cv2, cc1, cc2 = st
        H, q, ph = self.heads[si], self.qhd[si], self.phd[si]
        orig = src
        x = self.lin(src, p + "self_attn_weights.in_proj")
        T, B, _ = x.shape
        qd = q * H
        qq, kk, pp = x[..., :qd], x[..., qd : 2 * qd], x[..., 2 * qd :]
        kk = torch.cat([ck, kk], dim=0)
        ck = kk[-L:, ...]
        KL = kk.shape[0]
        qq = qq.reshape(T, B, H, q).permute(2, 1, 0, 3)
        pp = pp.reshape(T, B, H, ph).permute(2, 1, 0, 3)
        kk = kk.reshape(KL, B, H, q).permute(2, 1, 3, 0)
        scores = torch.matmul(qq, kk)
        pe = F.linear(pos_emb, w[p + "self_attn_weights.linear_pos.weight"])
        n2 = 2 * T - 1 + L
        pe = pe.reshape(-1, n2, H, ph).permute(2, 0, 3, 1)
        ps = torch.matmul(pp, pe)
        rows = torch.arange(T - 1, -1, -1).repeat(B * H).unsqueeze(-1)
        idx = rows + torch.arange(KL)
        ps = torch.gather(ps.reshape(-1, n2), 1, idx).reshape(H, B, T, KL)
        scores = (scores + ps).masked_fill(mask.unsqueeze(1), -1000)
        aw = scores.softmax(dim=-1)
        src = src + self.ff(p, 1, src)
        # NonlinAttention.streaming_forward
        y = self.lin(src, p + "nonlin_attention.in_proj")
        s_, x_, y_ = y.chunk(3, dim=2)
        x_ = (x_ * torch.tanh(s_)).reshape(T, B, 1, -1).permute(2, 1, 0, 3)
        xp = torch.cat([cn, x_], dim=2)
        cn = xp[:, :, -L:, :]
        x_ = torch.matmul(aw[0:1], xp).permute(2, 1, 0, 3).reshape(T, B, -1)
        src = src + self.lin(x_ * y_, p + "nonlin_attention.out_proj")

        def sa(k, src, cv):
            v = self.lin(src, p + f"self_attn{k}.in_proj")
            v = torch.cat([cv, v], dim=0)
            cv = v[-L:, ...]
            v = v.reshape(KL, B, H, -1).permute(2, 1, 0, 3)
            v = torch.matmul(aw, v).permute(2, 1, 0, 3).reshape(T, B, -1)
            return self.lin(v, p + f"self_attn{k}.out_proj"), cv

        a, cv1 = sa(1, src, cv1)
        src = src + a
        c, cc1 = self.conv_stream(si, p, 1, src, cc1)
        src = src + c
        src = src + self.ff(p, 2, src)
        src = orig + (src - orig) * w[p + "bypass_mid.bypass_scale"]
        a, cv2 = sa(2, src, cv2)
        src = src + a
        c, cc2 = self.conv_stream(si, p, 2, src, cc2)
        src = src + c
        src = src + self.ff(p, 3, src)
        src = bias_norm(src, w[p + "norm.bias"], w[p + "norm.log_scale"][0])
        return orig + (src - orig) * w[p + "bypass.bypass_scale"], [ck, cn, cv1, cv2, cc1, cc2]

    def encoder_chunk(self, x, states):
        """x: (N, T, 80) -> (N, T', J), new states   (OnnxEncoder.forward of the streaming export)"""
        N = x.size(0)
        emb, new_embed = self.embed_stream(x, states[-2])
        Tc = emb.size(1)
        left50 = self.left[0] * self.dss[0]
        pm = torch.arange(left50).expand(N, left50)
        pm = (states[-1].unsqueeze(1) <= pm).flip(1)
        new_len = states[-1] + Tc
        mask50 = torch.cat([pm, torch.zeros(N, Tc, dtype=torch.bool)], dim=1)
        x = emb.permute(1, 0, 2)
        outs, new_states, off = [], [], 0
        for si, D in enumerate(self.dims):
            ds, L = self.dss[si], self.left[si]
            c = x.shape[-1]
            x = x[..., :D] if D <= c else F.pad(x, (0, D - c))
            mask = mask50[..., ::ds]
            orig = x
            if ds > 1:
                x = self.downsample(x, self.w[f"encoder.encoders.{si}.downsample.bias"], ds)
            pe = self.pos_emb_stream(x.shape[0], L)
            for li in range(self.layers[si]):
                x, ns = self.layer_stream(si, li, x, pe, states[off * 6 : off * 6 + 6], L, mask)
                new_states += ns
                off += 1
            if ds > 1:
                T_, B_, D_ = x.shape
                x = x.unsqueeze(1).expand(T_, ds, B_, D_).reshape(T_ * ds, B_, D_)[: orig.shape[0]]
                x = orig + (x - orig) * self.w[f"encoder.encoders.{si}.out_combiner.bypass_scale"]
            outs.append(x)
        pieces, cur = [outs[-1]], self.dims[-1]
        for i in range(len(self.dims) - 2, -1, -1):
            if self.dims[i] > cur:
                pieces.append(outs[i][..., cur : self.dims[i]])
                cur = self.dims[i]
        x = torch.cat(pieces, dim=-1)
        x = self.downsample(x, self.w["encoder.downsample_output.bias"], 2)
        out = self.lin(x.permute(1, 0, 2), "joiner.encoder_proj")
        return out, new_states + [new_embed, new_len]
