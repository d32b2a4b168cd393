"""Repeat one deterministic training step (config 4: B = 16, T = 32000, 6 blocks, dropout 0.1) N times on one engine and compare
every output and parameter gradient bit for bit with the first run: a hazard or a race in a hand-scheduled kernel shows up as a run
that differs.   python3 tools/train_soak.py [N=60]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
eng = DptnEngine(DPTN_AV, dev)
eng.bind(params_to_device(synthetic_state_dict(DPTN_AV, 0), dev))
eng.bind_grads()
eng.set_option("deterministic", 1)
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(DPTN_AV, B=16, T=32000, Tv=50, seed=3).items()}
args = (t["mix"], t["s1_embedding"], t["s2_embedding"])
g = torch.Generator().manual_seed(9)
d1, d2 = (torch.randn(16, 32000, generator=g).to(dev) for _ in range(2))
first, bad = None, 0
for i in range(n):
    s1, s2, tape = eng.train_forward(*args)
    eng.train_backward(*args, d1, d2, tape)
    torch.cuda.synchronize()
    cur = {"s1": s1.clone(), "s2": s2.clone(), **{"grad." + k: x.clone() for k, x in eng._grads.items()}}
    del tape
    if first is None:
        first = cur
        assert all(torch.isfinite(v).all() for v in cur.values())
        continue
    diff = [k for k in cur if not torch.equal(cur[k], first[k])]
    if diff:
        bad += 1
        print(f"run {i}: {len(diff)} tensors differ, e.g. {diff[:3]}", flush=True)
print(f"{n} deterministic training steps, {bad} differ from the first")
sys.exit(1 if bad else 0)
