#!/usr/bin/env python3
"""GPU box: device time of the attention half of one TransformerDPRNN alone on the chip -- fused block (attn_block.hip)
vs the three separate launches (K1 + K2 + K3) -- at the bench shape.   python3 tools/attn_iso.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
S = eng.chunks(32000)
x = torch.randn(B, S, cfg.chunk_size, cfg.num_features, device=dev)
eng.set_option("overlap", 0)
for fuse in (1, 0):
    eng.set_option("fuse_attn", fuse)
    for path in (0, 1):
        eng.stage_path(0, path, x)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(5):
            eng.stage_path(0, path, x)
        prof = eng.profile_read()
        eng.profile(False)
        t = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
        part = t["attention"] + t.get("qkv_gemm", 0) + t.get("outproj_ln_gemm", 0)
        flops = B * S * cfg.chunk_size * (2 * 128 * 384 + 2 * 128 * 128 + 4 * (cfg.chunk_size if path == 0 else S) * 128)
        print(f"B={B} fuse={fuse} path={path}: attention half {part * 1e3:8.1f} us = {flops / part / 1e9:6.1f} TFLOP/s   "
              + " ".join(f"{k}={v * 1e3:.0f}" for k, v in t.items()))
