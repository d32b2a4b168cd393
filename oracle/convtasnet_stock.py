"""CPU ORACLE #3 (test infrastructure, NOT product code): the reference's Conv-TasNet (BASELINE configs[0], the
reference's own CPU-runnable case) composed from stock PyTorch operators, functional style.

Follows src/model/convtasnet.py: Encoder :6-15 (pad (16,32) + Conv1d(1,512,32,stride 16, no bias)), GlobalNorm :18-29
(mean/var over (C,T), eps 5e-6), Conv1D_Block :32-53 (1x1 conv -> PReLU -> GroupNorm(1,H,eps 1e-10) -> depthwise dilated
conv -> PReLU -> GroupNorm -> residual 1x1 / skip 1x1), Separator :55-83 (3 x 8 blocks, dilation 2^i, PReLU + 1x1 ->
sigmoid masks), Decoder :85-99 (ConvTranspose1d(512,1,32,stride 16) cropped [16, len-32)), ConvTasNet.forward :110-116.
No GPU work exists or is needed for this configuration; bench.py times it as an additional CPU datum.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

N, L, B, H, X, P, R = 512, 16, 128, 512, 8, 3, 3


def convtasnet_spec() -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict() keys and shapes of the reference's ConvTasNet() (5 066 929 parameters)."""
    out = [("encoder.conv1d.weight", (N, 1, 2 * L)), ("separator.norm_1.gamma", (N, 1)), ("separator.norm_1.beta", (N, 1)),
           ("separator.conv1d.weight", (B, N, 1)), ("separator.conv1d.bias", (B,))]
    for i in range(P * X):
        p = f"separator.separator.{i}."
        out += [(p + "conv1d.weight", (H, B, 1)), (p + "conv1d.bias", (H,)), (p + "PReLU_1.weight", (1,)),
                (p + "norm_1.weight", (H,)), (p + "norm_1.bias", (H,)), (p + "dconv1d.weight", (H, 1, R)),
                (p + "dconv1d.bias", (H,)), (p + "PReLU_2.weight", (1,)), (p + "norm_2.weight", (H,)),
                (p + "norm_2.bias", (H,)), (p + "conv.weight", (B, H, 1)), (p + "conv.bias", (B,)),
                (p + "conv_sc.weight", (B, H, 1)), (p + "conv_sc.bias", (B,))]
    out += [("separator.seq.0.weight", (1,)), ("separator.seq.1.weight", (2 * N, B, 1)), ("separator.seq.1.bias", (2 * N,)),
            ("decoder.deconv.weight", (N, 1, 2 * L))]
    return out


def synthetic_convtasnet_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    sd = {}
    for k, shape in convtasnet_spec():
        if k.endswith(("PReLU_1.weight", "PReLU_2.weight", "seq.0.weight")):
            w = np.full(shape, 0.25)
        elif k.endswith(("gamma", "norm_1.weight", "norm_2.weight")):
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif k.endswith(("beta", "norm_1.bias", "norm_2.bias")):
            w = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else H
            w = rng.uniform(-1, 1, size=shape) / np.sqrt(max(fan_in, 1))
        sd[k] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], mix: torch.Tensor) -> Dict[str, torch.Tensor]:
    bs = mix.shape[0]
    x = F.conv1d(F.pad(mix.unsqueeze(1), (L, 2 * L)), sd["encoder.conv1d.weight"], stride=L)
    enc = x
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    x = sd["separator.norm_1.gamma"] * (x - mu) / torch.sqrt(var + 5e-6) + sd["separator.norm_1.beta"]
    x = F.conv1d(x, sd["separator.conv1d.weight"], sd["separator.conv1d.bias"])
    acc = 0.0
    for i in range(P * X):
        p, dil = f"separator.separator.{i}.", 2 ** (i % X)
        c = F.conv1d(x, sd[p + "conv1d.weight"], sd[p + "conv1d.bias"])
        c = F.group_norm(F.prelu(c, sd[p + "PReLU_1.weight"]), 1, sd[p + "norm_1.weight"], sd[p + "norm_1.bias"], eps=1e-10)
        c = F.conv1d(c, sd[p + "dconv1d.weight"], sd[p + "dconv1d.bias"], padding=(dil * (R - 1)) // 2, dilation=dil, groups=H)
        c = F.group_norm(F.prelu(c, sd[p + "PReLU_2.weight"]), 1, sd[p + "norm_2.weight"], sd[p + "norm_2.bias"], eps=1e-10)
        x = x + F.conv1d(c, sd[p + "conv.weight"], sd[p + "conv.bias"])
        acc = acc + F.conv1d(c, sd[p + "conv_sc.weight"], sd[p + "conv_sc.bias"])
    m = torch.sigmoid(F.conv1d(F.prelu(acc, sd["separator.seq.0.weight"]), sd["separator.seq.1.weight"], sd["separator.seq.1.bias"]))
    y = (enc.unsqueeze(1) * m.reshape(bs, 2, N, -1)).reshape(-1, N, enc.shape[-1])
    y = F.conv_transpose1d(y, sd["decoder.deconv.weight"], stride=L)
    y = y[:, :, L:y.shape[2] - 2 * L].reshape(bs, 2, -1)
    return {"s1_pred": y[:, 0], "s2_pred": y[:, 1]}
