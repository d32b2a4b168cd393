"""Soak run (GPU box): many repetitions of the same forward / training step, every result compared BIT FOR BIT with the
first one (forward; the training step's gradients to 1e-5 of the largest element: their summation order is not fixed) -- a
rare race (a missing barrier, an unordered stream) shows up as a differing repetition.
usage: soak.py [forward_reps] [train_reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, DPRNN_AV, DPTNConfig, synthetic_inputs, synthetic_state_dict  # noqa: E402

freps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
treps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
bad = 0
for name, cfg, B, T, reps in (("dptn_av bs=1 (lstm4)", DPTN_AV, 1, 32000, freps), ("dptn_av B=3 (lstm4, whole batch)", DPTN_AV, 3, 32000, freps // 2),
                              ("dptn_av B=8 (lstm4, 2 sub-batches)", DPTN_AV, 8, 32000, freps // 4), ("dptn_av B=16", DPTN_AV, 16, 32000, freps // 5),
                              ("dptn_audio B=16", DPTN_AUDIO, 16, 32000, freps // 5),
                              ("dprnn_av B=4 x 2 s", DPTNConfig(**{**DPRNN_AV.to_dict(), "num_blocks": 2}), 4, 32000, freps // 5)):
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
    t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=7).items()}
    args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    ref = [x.clone() for x in eng.forward(*args)]
    t0 = time.time()
    diff = 0
    for i in range(reps):
        out = eng.forward(*args)
        if not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
            diff += 1
    torch.cuda.synchronize()
    print(f"{name}: {reps} forwards, {diff} differ from the first ({time.time() - t0:.1f} s)", flush=True)
    bad += diff
    del eng
    torch.cuda.empty_cache()

from speech_separation_amd import DPTNAVWavEncDec  # noqa: E402
from speech_separation_amd.train import SiSNRWavLoss  # noqa: E402
cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2, "dropout": 0.1})
kw = {k: v for k, v in cfg.to_dict().items() if k not in ("audio_only", "arch")}
model = DPTNAVWavEncDec(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
model = model.to(dev).train()
batch0 = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=6, T=16000, Tv=25, seed=9).items()}
crit = SiSNRWavLoss()
ref = None
diff = 0
worst = 0.0
t0 = time.time()
for i in range(treps):
    if i:
        model.zero_grad(set_to_none=True)       # as optimizer.zero_grad() in train_step: kept gradients would ACCUMULATE
    model._drop_step = 0                                          # same dropout masks every repetition (model._arm_dropout)
    batch = dict(batch0)
    batch.update(model(**batch))
    crit(**batch)["loss"].backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])      # (the flat buffer's alignment padding is not part of it)
    if ref is None:
        ref = g
    else:
        # gradients are reproducible up to fp32 summation order only: the token reductions of the weight gradients hand
        # their tiles out by dynamic tickets, so WHICH workgroup sums which tiles varies run to run
        worst = max(worst, float((g - ref).abs().max() / ref.abs().max()))
        if not bool(torch.isfinite(g).all()) or float((g - ref).abs().max()) > 1e-5 * float(ref.abs().max()):
            diff += 1
torch.cuda.synchronize()
print(f"training step (2 blocks, B=6, dropout 0.1): {treps} forward+backward passes, {diff} gradients further than 1e-5 (relative to the "
      f"largest element) from the first; worst {worst:.2e} ({time.time() - t0:.1f} s)", flush=True)
bad += diff
sys.exit(1 if bad else 0)
