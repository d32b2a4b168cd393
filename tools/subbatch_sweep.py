#!/usr/bin/env python3
"""GPU box: forward time of the bench workload (B = 16, T = 32000) against the number of sub-batches dptnav_forward cuts
the batch into (option sub_batches; 0 = forward_split's own choice), fp32 default and the opt-in split mode.
   python3 tools/subbatch_sweep.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=0).items()}
args = (inp["mix"], inp["s1_embedding"], inp["s2_embedding"])
ref = {}
for split in (0, 1):
    eng.set_option("split_bf16", split)
    for nsub in (0, 2, 3, 4, 6, 8):
        eng.set_option("sub_batches", nsub)
        for _ in range(3):
            out = eng.forward(*args)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            out = eng.forward(*args)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 10
        same = ""
        if nsub == 0:
            ref[split] = out[0].clone()
        else:
            same = "bit-identical to auto" if torch.equal(out[0], ref[split]) else "DIFFERS from auto"
        print(f"B={B} split={split} sub_batches={nsub}: {ms:7.2f} ms/step  {B / ms * 1e3:7.1f} mixtures/s  {same}", flush=True)
