#!/usr/bin/env python3
"""Developer probe (GPU box): stage-by-stage agreement of libdptnav with the numpy oracle.
Prints dB agreement per stage; not part of the product path."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dptn_oracle as O  # noqa: E402
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, DPTNConfig, synthetic_inputs, synthetic_state_dict  # noqa: E402


def db(a, b):
    return O.agreement_db(a, b)


def check(cfg: DPTNConfig, B, T, Tv, label):
    dev = torch.device("cuda:0")
    sd = synthetic_state_dict(cfg, seed=0)
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=123)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    taps = {}
    ref = O.forward(cfg, sd, dtype=np.float64, taps=taps, **inp)
    N, K = cfg.num_features, cfg.chunk_size
    e1, e2 = (t.get("s1_embedding"), t.get("s2_embedding"))
    enc, chk = eng.stage_head(t["mix"], e1, e2)
    torch.cuda.synchronize()
    print(f"[{label}] head: encoded {db(enc.cpu().numpy(), taps['encoded'].transpose(0, 2, 1)):.1f} dB, "
          f"chunked {db(chk.cpu().numpy(), taps['chunked'].transpose(0, 2, 3, 1)):.1f} dB")
    S = chk.shape[1]
    x = chk
    for b in range(cfg.num_blocks):
        for path, nm in ((0, "intra"), (1, "inter")):
            y = eng.stage_path(b, path, x)
            torch.cuda.synchronize()
            want = taps[f"blk{b}_{nm}"]
            if path == 0:
                want = want.reshape(B, S, K, N)
            else:
                want = want.reshape(B, K, S, N).transpose(0, 2, 1, 3)
            print(f"[{label}] blk{b} {nm}: {db(y.cpu().numpy(), want):.1f} dB  (nan={bool(torch.isnan(y).any())})")
            x = y
    s1, s2 = eng.stage_tail(x, enc, T)
    torch.cuda.synchronize()
    print(f"[{label}] tail: s1 {db(s1.cpu().numpy(), ref['s1_pred']):.1f} dB, s2 {db(s2.cpu().numpy(), ref['s2_pred']):.1f} dB")
    t0 = time.perf_counter()
    f1, f2 = eng.forward(t["mix"], e1, e2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{label}] forward: s1 {db(f1.cpu().numpy(), ref['s1_pred']):.1f} dB, s2 {db(f2.cpu().numpy(), ref['s2_pred']):.1f} dB "
          f"({dt * 1e3:.1f} ms first call)")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "av"
    if which in ("av", "all"):
        check(DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2}), B=2, T=8000, Tv=50, label="mid_av")
    if which in ("audio", "all"):
        check(DPTNConfig(**{**DPTN_AUDIO.to_dict(), "num_blocks": 2}), B=2, T=8000, Tv=50, label="mid_audio")
    if which in ("ragged", "all"):
        check(DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 1}), B=3, T=1000, Tv=7, label="ragged")
