"""Independent torch restatement of the LSTM transducer encoder (icefall lstm_transducer_stateless2) for cross-checking
oracle/k2_oracle_lstm.c: F.conv2d for Conv2dSubsampling, torch.nn.LSTM(proj_size=...) for the recurrence.  Shares no code with
the C oracle; not the reference, never shipped."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from torch_twin_conformer import basic_norm, double_swish


class LstmTwin:
    def __init__(self, meta: dict, tensors: dict):
        self.w = {k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
        self.D = int(meta["d_model"])
        self.Hh = int(meta["rnn_hidden_size"])
        self.L = int(meta["num_encoder_layers"])
        self.lstms = []
        for i in range(self.L):
            p = f"encoder.encoder.layers.{i}.lstm."
            m = torch.nn.LSTM(input_size=self.D, hidden_size=self.Hh, proj_size=self.D)
            with torch.no_grad():
                for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_hr_l0"):
                    getattr(m, n).copy_(self.w[p + n])
            self.lstms.append(m.eval())

    def embed(self, x):  # (N, T, 80) -> (N, T', D)
        w, e = self.w, "encoder.encoder_embed."
        x = x.unsqueeze(1)
        x = double_swish(F.conv2d(x, w[e + "conv.0.weight"], w[e + "conv.0.bias"], padding=(0, 1)))
        x = double_swish(F.conv2d(x, w[e + "conv.3.weight"], w[e + "conv.3.bias"], stride=2))
        x = double_swish(F.conv2d(x, w[e + "conv.6.weight"], w[e + "conv.6.bias"], stride=2))
        b, c, t, f = x.size()
        x = F.linear(x.transpose(1, 2).contiguous().view(b, t, c * f), w[e + "out.weight"], w[e + "out.bias"])
        return basic_norm(x, w[e + "out_norm.eps"])

    def layer(self, i, src, states):  # src (T, N, D); states (h (1,N,D), c (1,N,Hh))
        w, p = self.w, f"encoder.encoder.layers.{i}."
        y, new_states = self.lstms[i](src, states)
        src = y + src
        ff = F.linear(double_swish(F.linear(src, w[p + "feed_forward.0.weight"], w[p + "feed_forward.0.bias"])),
                      w[p + "feed_forward.4.weight"], w[p + "feed_forward.4.bias"])
        src = src + ff
        return basic_norm(src, w[p + "norm_final.eps"]), new_states

    @torch.no_grad()
    def forward(self, x, states=None, tap: int = -1):
        """x (N, T, 80).  Returns (out, new_states); states = (h (L,N,D), c (L,N,Hh)) or None for zeros."""
        x = self.embed(torch.as_tensor(x))
        N = x.size(0)
        if tap == 0:
            return x.numpy(), states
        if states is None:
            states = (torch.zeros(self.L, N, self.D), torch.zeros(self.L, N, self.Hh))
        x = x.permute(1, 0, 2)
        hs, cs = [], []
        for i in range(self.L):
            x, (h, c) = self.layer(i, x, (states[0][i : i + 1].contiguous(), states[1][i : i + 1].contiguous()))
            hs.append(h)
            cs.append(c)
            if tap == 1 + i:
                return x.permute(1, 0, 2).contiguous().numpy(), None
        x = x.permute(1, 0, 2)
        out = F.linear(x, self.w["joiner.encoder_proj.weight"], self.w["joiner.encoder_proj.bias"])
        return out.contiguous().numpy(), (torch.cat(hs, 0), torch.cat(cs, 0))
