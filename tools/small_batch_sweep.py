#!/usr/bin/env python3
"""GPU box: forward time of DPTN-AV (T = 32000) at small batches against the choices dptnav_forward makes for them --
recurrence kernel (lstm4 on 4-sequence tiles / lstm16 + K4 / the fused lstm16x128), number of sub-batches -- to set the
thresholds of forward_split / fuse128_for / use4.   python3 tools/small_batch_sweep.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

Bs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 10, 12]
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
# (label, options): lstm4 0 = never / 1 = by size / 2 = always; fuse_pre128 0 = never (K4 + lstm16) / 2 = always (lstm16x128)
KERNELS = [("auto", {"lstm4": 1, "fuse_pre128": 1}), ("lstm4", {"lstm4": 2, "fuse_pre128": 0}), ("K4+lstm16", {"lstm4": 0, "fuse_pre128": 0}),
           ("lstm16x128", {"lstm4": 0, "fuse_pre128": 2})]
for B in Bs:
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=0).items()}
    args = (inp["mix"], inp["s1_embedding"], inp["s2_embedding"])
    row = []
    for label, opts in KERNELS:
        for k, v in opts.items():
            eng.set_option(k, v)
        for nsub in (0, 1, 2, 3):
            if nsub > B:
                continue
            eng.set_option("sub_batches", nsub)
            for _ in range(3):
                eng.forward(*args)
            torch.cuda.synchronize()
            reps = 20 if B <= 4 else 10
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(reps):
                eng.forward(*args)
            ev1.record()
            torch.cuda.synchronize()
            row.append((ev0.elapsed_time(ev1) / reps, label, nsub))
    best = min(row)
    auto = [r for r in row if r[1] == "auto" and r[2] == 0][0]
    print(f"B={B:2d}: auto {auto[0]:6.3f} ms ({B / auto[0] * 1e3:6.1f} mixtures/s)   best {best[0]:6.3f} ms = {best[1]} / sub_batches={best[2]}   | "
          + "  ".join(f"{lab}/{ns}:{ms:.2f}" for ms, lab, ns in row), flush=True)
eng.set_option("sub_batches", 0)
