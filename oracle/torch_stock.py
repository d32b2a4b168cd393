"""CPU ORACLE #2 (test infrastructure, NOT product code): the same path composed from STOCK PyTorch
CPU operators -- the operators the reference's modules dispatch to (nn.Conv1d, nn.MultiheadAttention
fast path, nn.LSTM/oneDNN, F.unfold/F.fold, F.interpolate, nn.ConvTranspose1d).

Purpose: the ``cpu_baseline`` leg of bench.py.  The reference's Python cannot travel to the GPU box, so the
"reference CPU path" timed there is this composition (kind = "port"); tests/test_oracle_golden.py proves it
equal to the reference on the committed fixtures.  Only tests/, smoke() and bench.py's cpu_baseline may
import it; the product (speech_separation_amd/) never does.

Follows: src/model/dptn_wav.py:171-194 (forward), :35-61 (DPTNWav), src/model/dptn.py:36-52,62-79,
src/model/dprnn.py:122-136,145-163.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn


class StockDPTN:
    """Callable: (mix, s1_embedding, s2_embedding) -> {"s1_pred","s2_pred"} on CPU, fp32, eval/no_grad."""

    def __init__(self, cfg, state_dict: Dict[str, "object"]):
        self.cfg = cfg
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)).float() for k, v in state_dict.items()}
        self.sd = sd
        N, H = cfg.num_features, cfg.hidden_dim
        self.paths = []
        for b in range(cfg.num_blocks):
            for name in ("intra_chunk_block", "inter_chunk_block"):
                pre = f"dprnn.model.{b}.{name}."
                two = name == "intra_chunk_block" or cfg.bidir
                rnn = nn.LSTM(N, H, bidirectional=two, batch_first=True)
                rnn.load_state_dict({k[len(pre) + 4:]: v for k, v in sd.items() if k.startswith(pre + "rnn.")})
                mha = None
                if cfg.arch == "dptn":
                    mha = nn.MultiheadAttention(N, cfg.num_heads, dropout=cfg.dropout, batch_first=True)
                    mha.load_state_dict({k[len(pre) + 4:]: v for k, v in sd.items() if k.startswith(pre + "mha.")})
                    mha.eval()
                self.paths.append((pre, mha, rnn.eval()))

    def _path(self, x, pre, mha, rnn):
        sd, N = self.sd, self.cfg.num_features
        if mha is None:   # DPRNN: dprnn.py:24-47,65-89
            y = F.linear(rnn(x)[0], sd[pre + "fc.weight"], sd[pre + "fc.bias"])
            return F.layer_norm(y, (N,), sd[pre + "norm1d.weight"], sd[pre + "norm1d.bias"]) + x
        y = mha(x, x, x, need_weights=False)[0] + x
        y = F.layer_norm(y, (N,), sd[pre + "ln1.weight"], sd[pre + "ln1.bias"])
        r = rnn(y)[0]
        z = F.linear(F.relu(r), sd[pre + "ffn.1.weight"], sd[pre + "ffn.1.bias"]) + y
        return F.layer_norm(z, (N,), sd[pre + "ln2.weight"], sd[pre + "ln2.bias"])

    @torch.no_grad()
    def __call__(self, mix, s1_embedding=None, s2_embedding=None, **_):
        cfg, sd = self.cfg, self.sd
        N, K, P = cfg.num_features, cfg.chunk_size, cfg.step_size
        B, T = mix.shape
        enc = F.conv1d(mix.unsqueeze(1), sd["encoder.weight"], stride=cfg.stride_enc)
        L = enc.shape[-1]
        if not cfg.audio_only:
            W, b = sd["visual_compression.weight"], sd["visual_compression.bias"]
            video = torch.cat([F.linear(s1_embedding.permute(0, 2, 1), W, b),
                               F.linear(s2_embedding.permute(0, 2, 1), W, b)], -1)
            video = F.interpolate(video.permute(0, 2, 1), size=L, mode="linear", align_corners=False).permute(0, 2, 1)
            video = F.layer_norm(video, (N,), sd["video_ln.weight"], sd["video_ln.bias"])
            enc = enc + sd["gate"].tanh() * video.permute(0, 2, 1)
        x = F.unfold(enc.unsqueeze(-1), kernel_size=(K, 1), stride=(P, 1)).view(B, N, K, -1).permute(0, 1, 3, 2)
        S = x.shape[2]
        x = x.permute(0, 2, 3, 1).reshape(B * S, K, N)
        for i, (pre, mha, rnn) in enumerate(self.paths):
            x = self._path(x, pre, mha, rnn)
            if i % 2 == 0:   # (b s) k n -> (b k) s n
                x = x.view(B, S, K, N).transpose(1, 2).reshape(B * K, S, N)
            else:            # (b k) s n -> (b s) k n
                x = x.view(B, K, S, N).transpose(1, 2).reshape(B * S, K, N)
        x = x.view(B, S, K, N).permute(0, 3, 1, 2)                                   # b n s k
        x = F.prelu(x, sd["dprnn.speakers_separation.0.weight"])
        x = F.conv2d(x, sd["dprnn.speakers_separation.1.weight"], sd["dprnn.speakers_separation.1.bias"])
        ola = (S - 1) * P + K
        x = F.fold(x.permute(0, 1, 3, 2).reshape(B, 2 * N * K, S), output_size=(ola, 1), kernel_size=(K, 1),
                   stride=(P, 1)).squeeze(3)
        pad = L - ola
        x = F.pad(x, (pad // 2, pad - pad // 2)).view(B, 2, N, L).transpose(0, 1)
        preds = []
        for j in range(2):
            m = F.conv1d(x[j], sd["dprnn.postprocessing.0.weight"], sd["dprnn.postprocessing.0.bias"])
            y = F.conv_transpose1d(m + enc, sd["decoder.weight"], stride=cfg.stride_enc)
            padn = T - y.shape[-1]
            preds.append(F.pad(y, (padn // 2, padn - padn // 2)).squeeze(1))
        return {"s1_pred": preds[0], "s2_pred": preds[1]}


class SiSNRWavLossTorch(nn.Module):
    """The reference's training loss restated with the same PyTorch operators -- ORACLE for dptnav_pit_sisnr_loss and for
    torch.autograd comparisons of the training step (tests only).  Follows src/loss/ss_losses.py:100-114 (SiSNRLoss:
    zero-mean, projection on the target, -20 log10(|a t|^2 / |p - a t|^2), batch mean, no eps) and :21-26 (BaseSSLoss:
    batch-level PIT -- the smaller of the two permutations' batch means).  Pinned against values the reference's own
    SiSNRWavLoss produced (tests/golden/*.npz `val.pit_loss`) by tests/test_oracle_golden.py."""

    @staticmethod
    def pair(pred, gt):
        pred = pred - pred.mean(-1, keepdim=True)
        gt = gt - gt.mean(-1, keepdim=True)
        scale = (gt * pred).sum(-1, keepdim=True) / (gt * gt).sum(-1, keepdim=True)
        st = scale * gt
        return (-20 * torch.log10((st * st).sum(-1) / ((pred - st) ** 2).sum(-1))).mean()

    def forward(self, s1_pred, s2_pred, s1, s2, **batch):
        p1 = (self.pair(s1_pred, s1) + self.pair(s2_pred, s2)) / 2
        p2 = (self.pair(s1_pred, s2) + self.pair(s2_pred, s1)) / 2
        return {"loss": p2 if p2 < p1 else p1}
