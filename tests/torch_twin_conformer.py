"""Independent torch restatement of the offline Conformer encoder (icefall
pruned_transducer_stateless2 inference graph) for cross-checking oracle/k2_oracle_conformer.c.

Library ops throughout (F.conv2d / F.conv1d / F.glu / torch.matmul / as_strided rel-shift) in icefall's
(T, B, D) layout; shares no code with the C oracle.  Not the reference, never shipped.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def double_swish(x):
    return x * torch.sigmoid(x - 1.0)


def basic_norm(x, log_eps):
    scales = (torch.mean(x**2, dim=-1, keepdim=True) + log_eps.exp()) ** -0.5
    return x * scales


def rel_pos_encoding(T: int, D: int) -> torch.Tensor:
    """RelPositionalEncoding.extend_pe: [1, 2T-1, D], float32 arithmetic as torch does it."""
    pe_positive = torch.zeros(T, D)
    pe_negative = torch.zeros(T, D)
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
    pe_positive[:, 0::2] = torch.sin(position * div_term)
    pe_positive[:, 1::2] = torch.cos(position * div_term)
    pe_negative[:, 0::2] = torch.sin(-1 * position * div_term)
    pe_negative[:, 1::2] = torch.cos(-1 * position * div_term)
    pe_positive = torch.flip(pe_positive, [0]).unsqueeze(0)
    pe_negative = pe_negative[1:].unsqueeze(0)
    return torch.cat([pe_positive, pe_negative], dim=1)


def rel_shift(x):
    """(batch, head, time1, 2*time1-1) -> (batch, head, time1, time1) via as_strided, as icefall does."""
    (batch_size, num_heads, time1, n) = x.shape
    assert n == 2 * time1 - 1
    batch_stride, head_stride, time1_stride, n_stride = x.stride()
    return x.as_strided(
        (batch_size, num_heads, time1, time1),
        (batch_stride, head_stride, time1_stride - n_stride, n_stride),
        storage_offset=n_stride * (time1 - 1),
    )


class ConformerTwin:
    def __init__(self, meta: dict, tensors: dict):
        self.meta = meta
        self.w = {k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
        self.D = int(meta["encoder_dims"])
        self.L = int(meta["num_encoder_layers"])
        self.H = int(meta["num_heads"])
        self.K = int(meta["cnn_module_kernels"])

    def embed(self, x):  # x: (N, T, 80) -> (N, T', D)
        w = self.w
        e = "encoder.encoder_embed."
        x = x.unsqueeze(1)
        x = double_swish(F.conv2d(x, w[e + "conv.0.weight"], w[e + "conv.0.bias"], padding=1))
        x = double_swish(F.conv2d(x, w[e + "conv.3.weight"], w[e + "conv.3.bias"], stride=2))
        x = double_swish(F.conv2d(x, w[e + "conv.6.weight"], w[e + "conv.6.bias"], stride=2))
        b, c, t, f = x.size()
        x = F.linear(x.transpose(1, 2).contiguous().view(b, t, c * f), w[e + "out.weight"], w[e + "out.bias"])
        return basic_norm(x, w[e + "out_norm.eps"])

    def feed_forward(self, p, x):
        w = self.w
        h = double_swish(F.linear(x, w[p + ".0.weight"], w[p + ".0.bias"]))
        return F.linear(h, w[p + ".4.weight"], w[p + ".4.bias"])

    def self_attn(self, p, x, pos_emb):  # x: (T, B, D); pos_emb: (1, 2T-1, D)
        w = self.w
        T, B, D = x.shape
        H, hd = self.H, D // self.H
        scaling = float(hd) ** -0.5
        q, k, v = F.linear(x, w[p + "in_proj.weight"], w[p + "in_proj.bias"]).chunk(3, dim=-1)
        q = (q * scaling).contiguous().view(T, B, H, hd)
        k = k.contiguous().view(-1, B, H, hd)
        v = v.contiguous().view(-1, B * H, hd).transpose(0, 1)
        q = q.transpose(0, 1)  # (B, T, H, hd)
        pm = F.linear(pos_emb, w[p + "linear_pos.weight"]).view(1, -1, H, hd).transpose(1, 2)  # (1, H, 2T-1, hd)
        q_with_bias_u = (q + w[p + "pos_bias_u"]).transpose(1, 2)  # (B, H, T, hd)
        q_with_bias_v = (q + w[p + "pos_bias_v"]).transpose(1, 2)
        k = k.permute(1, 2, 3, 0)  # (B, H, hd, T)
        matrix_ac = torch.matmul(q_with_bias_u, k)
        matrix_bd = rel_shift(torch.matmul(q_with_bias_v, pm.transpose(-2, -1)).contiguous())
        aw = (matrix_ac + matrix_bd).view(B * H, T, -1)
        aw = F.softmax(aw, dim=-1)
        out = torch.bmm(aw, v)  # (B*H, T, hd)
        out = out.transpose(0, 1).contiguous().view(T, B, D)
        return F.linear(out, w[p + "out_proj.weight"], w[p + "out_proj.bias"])

    def conv_module(self, p, x):  # (T, B, D)
        w = self.w
        x = x.permute(1, 2, 0)  # (B, D, T)
        x = F.conv1d(x, w[p + "pointwise_conv1.weight"], w[p + "pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        x = F.conv1d(x, w[p + "depthwise_conv.weight"], w[p + "depthwise_conv.bias"], padding=(self.K - 1) // 2, groups=self.D)
        x = double_swish(x)
        x = F.conv1d(x, w[p + "pointwise_conv2.weight"], w[p + "pointwise_conv2.bias"])
        return x.permute(2, 0, 1)

    def layer(self, i, src, pos_emb):
        p = f"encoder.encoder.layers.{i}."
        src = src + self.feed_forward(p + "feed_forward_macaron", src)
        src = src + self.self_attn(p + "self_attn.", src, pos_emb)
        src = src + self.conv_module(p + "conv_module.", src)
        src = src + self.feed_forward(p + "feed_forward", src)
        return basic_norm(src, self.w[p + "norm_final.eps"])

    @torch.no_grad()
    def forward(self, x, tap: int = -1):
        """x: (B, T, 80) float32.  tap 0: embed out (B,T',D); tap 1+i: after layer i; -1: encoder_out (B,T',J)."""
        x = self.embed(torch.as_tensor(x))
        if tap == 0:
            return x.numpy()
        pos_emb = rel_pos_encoding(x.size(1), self.D)
        x = x.permute(1, 0, 2)
        for i in range(self.L):
            x = self.layer(i, x, pos_emb)
            if tap == 1 + i:
                return x.permute(1, 0, 2).contiguous().numpy()
        x = x.permute(1, 0, 2)
        out = F.linear(x, self.w["joiner.encoder_proj.weight"], self.w["joiner.encoder_proj.bias"])
        return out.contiguous().numpy()

    @torch.no_grad()
    def decoder(self, y):
        """Stateless decoder, groups = DD / conv.weight.shape[1]; id < 0 -> zero embedding."""
        w = self.w
        y = torch.as_tensor(y, dtype=torch.int64)
        emb = w["decoder.embedding.weight"][y.clamp(min=0)] * (y >= 0).unsqueeze(-1)
        cw = w["decoder.conv.weight"]
        groups = cw.shape[0] // cw.shape[1]
        h = F.relu(F.conv1d(emb.permute(0, 2, 1), cw, groups=groups).permute(0, 2, 1)).squeeze(1)
        return F.linear(h, w["joiner.decoder_proj.weight"], w["joiner.decoder_proj.bias"]).numpy()


# ------------------------------------------------------------------------------------------------ streaming (chunk_forward)
def rel_pos_encoding_left(T: int, left: int, D: int) -> torch.Tensor:
    """RelPositionalEncoding.forward(x, left_context): rows for relative positions (left + T - 1) .. -(T - 1)"""
    x_size_1 = T + left
    pe = rel_pos_encoding(x_size_1, D)          # [1, 2*x_size_1 - 1, D], centre at index x_size_1 - 1
    center = pe.size(1) // 2
    return pe[:, center - x_size_1 + 1 : center + T]


def rel_shift_left(x, left_context: int):
    (batch_size, num_heads, time1, n) = x.shape
    time2 = time1 + left_context
    assert n == left_context + 2 * time1 - 1
    batch_stride, head_stride, time1_stride, n_stride = x.stride()
    return x.as_strided((batch_size, num_heads, time1, time2), (batch_stride, head_stride, time1_stride - n_stride, n_stride),
                        storage_offset=n_stride * (time1 - 1))


class ConformerStreamTwin(ConformerTwin):
    def __init__(self, meta, tensors):
        super().__init__(meta, tensors)
        self.left = int(meta["left_context"])
        self.right = int(meta.get("right_context", "0"))

    def init_states(self, N=1):
        return [torch.zeros(self.L, self.left, N, self.D), torch.zeros(self.L, self.K - 1, N, self.D)]

    def attn_chunk(self, p, src, key, pos_emb, key_padding_mask):
        w = self.w
        T, B, D = src.shape
        S = key.shape[0]
        H, hd = self.H, D // self.H
        W_, b_ = w[p + "in_proj.weight"], w[p + "in_proj.bias"]
        q = F.linear(src, W_[:D], b_[:D])
        k, v = F.linear(key, W_[D:], b_[D:]).chunk(2, dim=-1)
        q = (q * float(hd) ** -0.5).contiguous().view(T, B, H, hd).transpose(0, 1)
        k = k.contiguous().view(S, B, H, hd)
        v = v.contiguous().view(S, B * H, hd).transpose(0, 1)
        pm = F.linear(pos_emb, w[p + "linear_pos.weight"]).view(1, -1, H, hd).transpose(1, 2)
        qu = (q + w[p + "pos_bias_u"]).transpose(1, 2)
        qv = (q + w[p + "pos_bias_v"]).transpose(1, 2)
        ac = torch.matmul(qu, k.permute(1, 2, 3, 0))
        bd = rel_shift_left(torch.matmul(qv, pm.transpose(-2, -1)).contiguous(), self.left)
        aw = (ac + bd).view(B, H, T, S).masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2), float("-inf")).view(B * H, T, S)
        aw = F.softmax(aw, dim=-1)
        out = torch.bmm(aw, v).transpose(0, 1).contiguous().view(T, B, D)
        return F.linear(out, w[p + "out_proj.weight"], w[p + "out_proj.bias"])

    def conv_chunk(self, p, x, cache):  # x (T, B, D); cache (K-1, B, D)
        w = self.w
        x = x.permute(1, 2, 0)
        x = F.glu(F.conv1d(x, w[p + "pointwise_conv1.weight"], w[p + "pointwise_conv1.bias"]), dim=1)
        x = torch.cat([cache.permute(1, 2, 0), x], dim=2)
        cache = x.permute(2, 0, 1)   # (lorder + T, B, D)
        cache = cache[-(self.K - 1 + self.right): -self.right, ...] if self.right > 0 else cache[-(self.K - 1):, ...]
        x = F.conv1d(x, w[p + "depthwise_conv.weight"], w[p + "depthwise_conv.bias"], groups=self.D)   # causal: no padding
        x = F.conv1d(double_swish(x), w[p + "pointwise_conv2.weight"], w[p + "pointwise_conv2.bias"])
        return x.permute(2, 0, 1), cache

    @torch.no_grad()
    def chunk(self, x, states, processed_lens):
        """Conformer.streaming_forward (simulate_streaming=False): x (N, T, 80) -> (N, chunk, J), new states.  right_context > 0: the
        chunk's last `right` encoder frames are seen by this step and cut from its output; the caches keep what is in front of them."""
        x = torch.as_tensor(x)
        N = x.size(0)
        embed = self.embed(x)[:, 1:-1, :]
        T = embed.size(1)
        pos_emb = rel_pos_encoding_left(T, self.left, self.D)
        processed_mask = torch.arange(self.left).expand(N, self.left)
        processed_mask = (processed_lens.view(N, 1) <= processed_mask).flip(1)
        mask = torch.cat([processed_mask, torch.zeros(N, T, dtype=torch.bool)], dim=1)
        src = embed.permute(1, 0, 2)
        new_attn, new_conv = [], []
        for i in range(self.L):
            p = f"encoder.encoder.layers.{i}."
            src = src + self.feed_forward(p + "feed_forward_macaron", src)
            key = torch.cat([states[0][i], src], dim=0)
            new_attn.append(key[-(self.left + self.right): -self.right, ...] if self.right > 0 else key[-self.left:, ...])
            src = src + self.attn_chunk(p + "self_attn.", src, key, pos_emb, mask)
            conv, cc = self.conv_chunk(p + "conv_module.", src, states[1][i])
            new_conv.append(cc)
            src = src + conv
            src = src + self.feed_forward(p + "feed_forward", src)
            src = basic_norm(src, self.w[p + "norm_final.eps"])
        if self.right > 0:
            src = src[:-self.right, ...]
        out = F.linear(src.permute(1, 0, 2), self.w["joiner.encoder_proj.weight"], self.w["joiner.encoder_proj.bias"])
        return out.contiguous().numpy(), [torch.stack(new_attn), torch.stack(new_conv)], processed_lens + T - self.right
