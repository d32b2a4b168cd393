#!/usr/bin/env python3
"""GPU box: the training step (config 4, B = 16) against the scheduling options of dptnav_train_forward / _backward.
   python3 tools/train_options_sweep.py"""
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("AB_LIB"):          # A/B of two builds of the library: AB_LIB=<path to the other .so>
    import speech_separation_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])
from speech_separation_amd import DPTNAVWavEncDec  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402
from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step  # noqa: E402

dev = torch.device("cuda:0")
kw = {k: v for k, v in DPTN_AV.to_dict().items() if k not in ("audio_only", "arch")}
model = DPTNAVWavEncDec(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
model = model.to(dev).train()
opt = FusedAdamW(model.parameters(), lr=1e-3)
batch0 = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(DPTN_AV, B=16, T=32000, Tv=50, seed=123).items()}
crit = SiSNRWavLoss()
train_step(model, dict(batch0), crit, opt, 10.0)
eng = model._engine
defaults = {"wgrad_side": 1, "wgrad_ride": 1, "lstm_chain": 0, "ln_tape": 1, "wgrad2": 1, "lstm16": 1, "deterministic": 0}
cases = [{}] + [{k: 1 - v} for k, v in defaults.items()] + [{"wgrad_side": 0, "lstm_chain": 1}]
if len(sys.argv) > 1 and sys.argv[1] == "pack":
    defaults["pack_wih"] = 1
    cases = [{}, {"pack_wih": 0}, {}, {"pack_wih": 0}, {}, {"pack_wih": 0}]
if len(sys.argv) > 1 and sys.argv[1] == "det":
    cases = [{}, {"deterministic": 1}, {}, {"deterministic": 1}]
if len(sys.argv) > 1 and sys.argv[1] == "side":
    cases = [{}, {"wgrad_side": 0}, {"wgrad_ride": 0}, {"wgrad_side": 0, "wgrad_ride": 0}, {}, {"wgrad_side": 0}, {"wgrad_ride": 0},
             {"wgrad_side": 0, "wgrad_ride": 0}, {}]
for case in cases:
    for k, v in {**defaults, **case}.items():
        eng.set_option(k, v)
    for _ in range(3):
        train_step(model, dict(batch0), crit, opt, 10.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        train_step(model, dict(batch0), crit, opt, 10.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"{case or 'defaults'}: {1e3 * dt:.2f} ms/step {16 / dt:.1f} mixtures/s", flush=True)
