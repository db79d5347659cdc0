"""Independent torch restatement of the streaming Zipformer (v1) encoder chunk -- a second opinion for the C oracle
(oracle/k2_oracle_zipformer1.c).  Written module by module after icefall's pruned_transducer_stateless7_streaming
zipformer.py (*.streaming_forward), in torch's own tensor vocabulary (cumsum, as_strided rel-shift, conv1d / conv2d, bmm), so
that an indexing slip in the C loops cannot cancel out.  States use the reference's batch-1 tensor shapes
(OnlineProjOfZipformer.cs:430-487): cached_len [L,1], cached_avg [L,1,D], cached_key [L,left,1,A], cached_val / cached_val2
[L,left,1,A/2], cached_conv1 / cached_conv2 [L,1,D,K-1]."""
import math

import torch
import torch.nn.functional as F


def _ints(meta, k):
    return [int(x) for x in meta[k].split(",") if x]


def dswish(x):
    return x * torch.sigmoid(x - 1.0)


class Zipformer1Twin:
    def __init__(self, meta, tensors):
        self.meta = meta
        self.w = {k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
        self.dims = _ints(meta, "encoder_dims")
        self.atts = _ints(meta, "attention_dims")
        self.layers = _ints(meta, "num_encoder_layers")
        self.heads = _ints(meta, "num_heads")
        self.kern = _ints(meta, "cnn_module_kernels")
        self.ds = _ints(meta, "downsampling_factors")
        self.left = _ints(meta, "left_context_len") if "left_context_len" in meta else []
        self.pos_dim = int(meta["pos_dim"])
        self.T = int(meta.get("T", 0))

    def init_states(self):
        st = []
        for D, A, L, K, left in zip(self.dims, self.atts, self.layers, self.kern, self.left):
            st.append(dict(len=torch.zeros(L, 1), avg=torch.zeros(L, 1, D), key=torch.zeros(L, left, 1, A), val=torch.zeros(L, left, 1, A // 2),
                           val2=torch.zeros(L, left, 1, A // 2), conv1=torch.zeros(L, 1, D, K - 1), conv2=torch.zeros(L, 1, D, K - 1)))
        return st

    # ---- modules ----
    def embed(self, x):  # [N,T,80] -> [N,(T-7)//2,D0]
        w = self.w
        e = "encoder.encoder_embed."
        x = x.unsqueeze(1)
        x = dswish(F.conv2d(x, w[e + "conv.0.weight"], w[e + "conv.0.bias"], padding=(0, 1)))
        x = dswish(F.conv2d(x, w[e + "conv.3.weight"], w[e + "conv.3.bias"], stride=2))
        x = dswish(F.conv2d(x, w[e + "conv.6.weight"], w[e + "conv.6.bias"], stride=(1, 2)))
        b, c, t, f = x.shape
        return F.linear(x.transpose(1, 2).reshape(b, t, c * f), w[e + "out.weight"], w[e + "out.bias"])

    @staticmethod
    def rel_pos(T, left, D):
        # RelPositionalEncoding.extend_pe + forward(x, left_context_len)
        n = T + left
        position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
        pos_p = torch.zeros(n, D)
        neg_p = torch.zeros(n, D)
        pos_p[:, 0::2] = torch.sin(position * div)
        pos_p[:, 1::2] = torch.cos(position * div)
        neg_p[:, 0::2] = torch.sin(-1 * position * div)
        neg_p[:, 1::2] = torch.cos(-1 * position * div)
        pe = torch.cat([torch.flip(pos_p, [0]), neg_p[1:]], dim=0)
        center = pe.shape[0] // 2
        return pe[center - n + 1 : center + T]  # [left + 2T - 1, D]

    def attn_downsample(self, pfx, src, ds):  # [T,N,Din] -> [ceil(T/ds),N,Dout]
        T, N, Din = src.shape
        Td = (T + ds - 1) // ds
        if T != Td * ds:
            src = torch.cat([src, src[-1:].expand(Td * ds - T, N, Din)], dim=0)
        src = src.reshape(Td, ds, N, Din)
        scores = (src * self.w[pfx + "query"]).sum(dim=-1, keepdim=True)
        ans = (src * scores.softmax(dim=1)).sum(dim=1)
        if pfx + "extra_proj.weight" in self.w:
            flat = src.permute(0, 2, 1, 3).reshape(Td, N, ds * Din)
            ans = torch.cat([ans, F.linear(flat, self.w[pfx + "extra_proj.weight"])], dim=2)
        return ans

    @staticmethod
    def combine(s1, s2, w1):
        s1 = s1 * w1
        s2 = s2 * (1.0 - w1)
        d1, d2 = s1.shape[-1], s2.shape[-1]
        if d1 < d2:
            s1 = F.pad(s1, (0, d2 - d1))
        elif d1 > d2:
            s1 = s1[..., :d2]
        return s1 + s2

    def ff(self, p, k, x):
        w = self.w
        h = dswish(F.linear(x, w[p + f"feed_forward{k}.in_proj.weight"], w[p + f"feed_forward{k}.in_proj.bias"]))
        return F.linear(h, w[p + f"feed_forward{k}.out_proj.weight"], w[p + f"feed_forward{k}.out_proj.bias"])

    def conv(self, p, k, x, cache):  # x [T,N,D], cache [N,D,K-1]
        w = self.w
        q = p + f"conv_module{k}."
        x = x.permute(1, 2, 0)
        x = F.glu(F.conv1d(x, w[q + "pointwise_conv1.weight"], w[q + "pointwise_conv1.bias"]), dim=1)
        lo = cache.shape[2]
        x = torch.cat([cache, x], dim=2)
        cache = x[:, :, -lo:]
        x = dswish(F.conv1d(x, w[q + "depthwise_conv.weight"], w[q + "depthwise_conv.bias"], groups=x.shape[1]))
        x = F.conv1d(x, w[q + "pointwise_conv2.weight"], w[q + "pointwise_conv2.bias"])
        return x.permute(2, 0, 1), cache

    def layer(self, p, si, src, pos_emb, st, li):
        w = self.w
        A, H, P = self.atts[si], self.heads[si], self.pos_dim
        hd = A // H
        orig = src
        src = src + self.ff(p, 1, src)
        # pooling
        clen, cavg = st["len"][li], st["avg"][li]
        x = src.cumsum(dim=0) + (cavg * clen.unsqueeze(1)).unsqueeze(0)
        cum = torch.arange(1, x.shape[0] + 1).unsqueeze(1) + clen.unsqueeze(0)
        x = x * (1.0 / cum).unsqueeze(2)
        st["len"][li] = clen + x.shape[0]
        st["avg"][li] = x[-1]
        src = src + F.linear(x, w[p + "pooling.proj.weight"])
        # attention
        T, N, _ = src.shape
        xp = F.linear(src, w[p + "self_attn.in_proj.weight"], w[p + "self_attn.in_proj.bias"])
        pos = F.linear(pos_emb, w[p + "self_attn.linear_pos.weight"])
        q, k_, v, pq = xp[..., :A], xp[..., A : 2 * A], xp[..., 2 * A : 2 * A + A // 2], xp[..., 2 * A + A // 2 :]
        left = st["key"][li].shape[0]
        k_ = torch.cat([st["key"][li], k_], dim=0)
        v = torch.cat([st["val"][li], v], dim=0)
        st["key"][li] = k_[-left:]
        st["val"][li] = v[-left:]
        kv = k_.shape[0]
        q = q.reshape(T, N, H, hd).permute(1, 2, 0, 3)
        pq = pq.reshape(T, N, H, P).permute(1, 2, 0, 3)
        kk = k_.reshape(kv, N, H, hd).permute(1, 2, 3, 0)
        vv = v.reshape(kv, N * H, hd // 2).transpose(0, 1)
        T2 = 2 * T - 1 + left
        pos = pos.reshape(1, T2, H, P).permute(0, 2, 3, 1)
        pw = torch.matmul(pq, pos).contiguous()
        pw = pw.as_strided((N, H, T, kv), (pw.stride(0), pw.stride(1), pw.stride(2) - pw.stride(3), pw.stride(3)),
                           storage_offset=pw.stride(3) * (T - 1))
        aw = (torch.matmul(q, kk) + pw).view(N * H, T, kv).softmax(dim=-1)
        out = torch.bmm(aw, vv).transpose(0, 1).contiguous().view(T, N, A // 2)
        src = src + F.linear(out, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        c, st["conv1"][li] = self.conv(p, 1, src, st["conv1"][li])
        src = src + c
        src = src + self.ff(p, 2, src)
        v2 = F.linear(src, w[p + "self_attn.in_proj2.weight"])
        v2 = torch.cat([st["val2"][li], v2], dim=0)
        st["val2"][li] = v2[-left:]
        vv2 = v2.reshape(kv, N * H, hd // 2).transpose(0, 1)
        out = torch.bmm(aw, vv2).transpose(0, 1).contiguous().view(T, N, A // 2)
        src = src + F.linear(out, w[p + "self_attn.out_proj2.weight"], w[p + "self_attn.out_proj2.bias"])
        c, st["conv2"][li] = self.conv(p, 2, src, st["conv2"][li])
        src = src + c
        src = src + self.ff(p, 3, src)
        eps = w[p + "norm_final.eps"].exp()
        src = src * (src.pow(2).mean(dim=-1, keepdim=True) + eps) ** -0.5
        return orig + (src - orig) * w[p + "bypass_scale"]

    def skip_layer(self, i):
        z = self.ds
        if i <= 1 or z[i - 1] <= z[i]:
            return None
        for j in range(i - 2, -1, -1):
            if z[j] <= z[i] or j == 0:
                return j
        return None

    def chunk(self, x, states):
        """x [T,80] (log-floored) -> encoder_out [T',J] after joiner.encoder_proj; states updated in place."""
        w = self.w
        x = self.embed(x.unsqueeze(0)).permute(1, 0, 2)  # [T,1,D0]
        outputs = []
        for i, ds in enumerate(self.ds):
            k = self.skip_layer(i)
            if k is not None:
                x = self.combine(outputs[k], x, w[f"encoder.skip_modules.{i}.weight1"])
            st = states[i]
            base = f"encoder.encoders.{i}."
            if ds == 1:
                pe = self.rel_pos(x.shape[0], self.left[i], self.dims[i]).unsqueeze(0)
                for li in range(self.layers[i]):
                    x = self.layer(base + f"layers.{li}.", i, x, pe, st, li)
            else:
                orig = x
                xd = self.attn_downsample(base + "downsample.", x, ds)
                pe = self.rel_pos(xd.shape[0], self.left[i], self.dims[i]).unsqueeze(0)
                for li in range(self.layers[i]):
                    xd = self.layer(base + f"encoder.layers.{li}.", i, xd, pe, st, li)
                T, N, C = xd.shape
                up = (xd.unsqueeze(1).expand(T, ds, N, C) + w[base + "upsample.bias"].unsqueeze(1)).reshape(T * ds, N, C)
                x = self.combine(orig, up[: orig.shape[0]], w[base + "out_combiner.weight1"])
            outputs.append(x)
        x = self.attn_downsample("encoder.downsample_output.", x, 2).permute(1, 0, 2)[0]
        return F.linear(x, w["joiner.encoder_proj.weight"], w["joiner.encoder_proj.bias"])

    # ---- offline graph (pruned_transducer_stateless7 Zipformer.forward, x_lens = T) ----
    def conv_offline(self, p, k, x):  # x [T,N,D]
        w = self.w
        q = p + f"conv_module{k}."
        x = x.permute(1, 2, 0)
        x = F.glu(F.conv1d(x, w[q + "pointwise_conv1.weight"], w[q + "pointwise_conv1.bias"]), dim=1)
        K = w[q + "depthwise_conv.weight"].shape[2]
        x = dswish(F.conv1d(x, w[q + "depthwise_conv.weight"], w[q + "depthwise_conv.bias"], groups=x.shape[1], padding=K // 2))
        x = F.conv1d(x, w[q + "pointwise_conv2.weight"], w[q + "pointwise_conv2.bias"])
        return x.permute(2, 0, 1)

    def layer_offline(self, p, si, src, pos_emb):
        w = self.w
        A, H, P = self.atts[si], self.heads[si], self.pos_dim
        hd = A // H
        orig = src
        src = src + self.ff(p, 1, src)
        T, N, _ = src.shape
        # PoolingModule.forward with an all-False key_padding_mask: weight 1/T per frame, summed over time
        pooling_mask = torch.ones(N, T)
        pooling_mask = pooling_mask / pooling_mask.sum(dim=1, keepdim=True)
        pooling_mask = pooling_mask.transpose(0, 1).contiguous().unsqueeze(-1)
        pooled = (src * pooling_mask).sum(dim=0, keepdim=True)
        src = src + F.linear(pooled, w[p + "pooling.proj.weight"])
        xp = F.linear(src, w[p + "self_attn.in_proj.weight"], w[p + "self_attn.in_proj.bias"])
        pos = F.linear(pos_emb, w[p + "self_attn.linear_pos.weight"])
        q, k_, v, pq = xp[..., :A], xp[..., A : 2 * A], xp[..., 2 * A : 2 * A + A // 2], xp[..., 2 * A + A // 2 :]
        q = q.reshape(T, N, H, hd).permute(1, 2, 0, 3)
        pq = pq.reshape(T, N, H, P).permute(1, 2, 0, 3)
        kk = k_.reshape(T, N, H, hd).permute(1, 2, 3, 0)
        vv = v.reshape(T, N * H, hd // 2).transpose(0, 1)
        pos = pos.reshape(1, 2 * T - 1, H, P).permute(0, 2, 3, 1)
        pw = torch.matmul(pq, pos).contiguous()
        pw = pw.as_strided((N, H, T, T), (pw.stride(0), pw.stride(1), pw.stride(2) - pw.stride(3), pw.stride(3)), storage_offset=pw.stride(3) * (T - 1))
        aw = (torch.matmul(q, kk) + pw).view(N * H, T, T).softmax(dim=-1)
        out = torch.bmm(aw, vv).transpose(0, 1).contiguous().view(T, N, A // 2)
        src = src + F.linear(out, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        src = src + self.conv_offline(p, 1, src)
        src = src + self.ff(p, 2, src)
        v2 = F.linear(src, w[p + "self_attn.in_proj2.weight"])
        vv2 = v2.reshape(T, N * H, hd // 2).transpose(0, 1)
        out = torch.bmm(aw, vv2).transpose(0, 1).contiguous().view(T, N, A // 2)
        src = src + F.linear(out, w[p + "self_attn.out_proj2.weight"], w[p + "self_attn.out_proj2.bias"])
        src = src + self.conv_offline(p, 2, src)
        src = src + self.ff(p, 3, src)
        eps = w[p + "norm_final.eps"].exp()
        src = src * (src.pow(2).mean(dim=-1, keepdim=True) + eps) ** -0.5
        return orig + (src - orig) * w[p + "bypass_scale"]

    def forward_offline(self, x, tap=-1):
        """x [N,T,80] -> encoder_out [N,T',J] (after joiner.encoder_proj); tap: 0 embed output, 1+i output of stack i ([N,T50,D])."""
        w = self.w
        x = self.embed(x).permute(1, 0, 2)
        if tap == 0:
            return x.permute(1, 0, 2)
        outputs = []
        for i, ds in enumerate(self.ds):
            k = self.skip_layer(i)
            if k is not None:
                x = self.combine(outputs[k], x, w[f"encoder.skip_modules.{i}.weight1"])
            base = f"encoder.encoders.{i}."
            if ds == 1:
                pe = self.rel_pos(x.shape[0], 0, self.dims[i]).unsqueeze(0)
                for li in range(self.layers[i]):
                    x = self.layer_offline(base + f"layers.{li}.", i, x, pe)
            else:
                orig = x
                xd = self.attn_downsample(base + "downsample.", x, ds)
                pe = self.rel_pos(xd.shape[0], 0, self.dims[i]).unsqueeze(0)
                for li in range(self.layers[i]):
                    xd = self.layer_offline(base + f"encoder.layers.{li}.", i, xd, pe)
                T, N, C = xd.shape
                up = (xd.unsqueeze(1).expand(T, ds, N, C) + w[base + "upsample.bias"].unsqueeze(1)).reshape(T * ds, N, C)
                x = self.combine(orig, up[: orig.shape[0]], w[base + "out_combiner.weight1"])
            outputs.append(x)
            if tap == i + 1:
                return x.permute(1, 0, 2)
        x = self.attn_downsample("encoder.downsample_output.", x, 2).permute(1, 0, 2)
        return F.linear(x, w["joiner.encoder_proj.weight"], w["joiner.encoder_proj.bias"])
