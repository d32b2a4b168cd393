#!/usr/bin/env python3
"""Diagnostic (GPU box): per-segment cycle shares of one LSTM step, from the STAMP build of the kernel."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
# usage: lstm_stamps.py [B] [tile]   (tile 16 needs ceil(B*S/16)*2 <= CUs, i.e. B <= 8 at T = 32000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = 32000
S = eng.chunks(T)
eng.set_option("lstm16", 1 if tile == 16 else 0)
if len(sys.argv) > 3:
    eng.set_option("lstm_diag", int(sys.argv[3]))   # timing-only ablations (results are wrong)
x = torch.randn(B, S, cfg.chunk_size, cfg.num_features, device=dev)
eng.stage_path(0, 0, x)
eng.set_option("lstm_stamps", 1)
eng.stage_path(0, 0, x)
torch.cuda.synchronize()
raw = eng.tap("lstm_stamps", B, eng._path_T(S)).view(torch.int64).cpu().numpy()
nst = (B * S + tile - 1) // tile
a = raw[: 2 * nst * 4 * 4].reshape(2, nst, 4, 4).astype(np.float64) / cfg.chunk_size
print(f"B={B} tile={tile}");print("cycles per step (mean over waves):  acc-init %.0f   mfma %.0f   cell %.0f   barrier %.0f   total %.0f"
      % (*a.mean((0, 1, 2)), a.sum(-1).mean()))
print("per wave (tile 0, dir 0):", a[0, 0].round(0).tolist())
print("max-over-waves total:", a.sum(-1).max(), " min:", a.sum(-1).min())

# as run by dptnav_forward: two chained half-batches beside each other's GEMM / attention kernels.  The stamps of the
# LAST recurrence of half 0 (block 5, inter path) are left in that half's workspace slice (= the B/2 plan at offset 0).
if len(sys.argv) > 4 and sys.argv[4] == "asrun":
    from speech_separation_amd.spec import synthetic_inputs
    Bf = 2 * B
    inp = synthetic_inputs(cfg, B=Bf, T=T, Tv=50, seed=123)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    for _ in range(3):
        eng.forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
    eng.profile(True)
    eng.profile_reset()
    for _ in range(5):
        eng.forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
    prof = eng.profile_read()
    torch.cuda.synchronize()
    raw = eng.tap("lstm_stamps", B, T, 50).view(torch.int64).cpu().numpy()
    nst_i = (B * cfg.chunk_size + tile - 1) // tile
    a = raw[: 2 * nst_i * 4 * 4].reshape(2, nst_i, 4, 4).astype(np.float64) / S
    ms, n = prof["lstm_recurrence"]
    cyc = a.sum(-1).mean()
    print("AS RUN (B=%d forward, overlap): cycles per step %.0f  [init %.0f mfma %.0f cell %.0f barrier %.0f]; mean launch %.4f ms"
          % (Bf, cyc, *a.mean((0, 1, 2)), ms / n))
    print("  implied clock ~ %.2f GHz (cycles x mean steps / mean launch time)" % (cyc * (S + cfg.chunk_size) / 2 / (ms / n * 1e-3) / 1e9))
