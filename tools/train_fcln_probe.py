"""Training forward of DPTN-AV at B = 8 x 4 s (one sub-batch: every launch alone on the chip): device time of the out-projection +
LayerNorm 1 and FFN + LayerNorm 2 launches (with the LayerNorm tape) by fcln.hip (option fcln = 1) and by the GEMM engine (0).
python3 tools/train_fcln_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
eng.bind_grads()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=1).items()}
args = (t["mix"], t["s1_embedding"], t["s2_embedding"])
for rnd in range(2):
    for v in (0, 1):
        eng.set_option("fcln", v)
        for _ in range(2):
            s1, s2, tape = eng.train_forward(*args)
            del tape
        eng.profile(True)
        eng.profile_reset()
        for _ in range(3):
            s1, s2, tape = eng.train_forward(*args)
            del tape
        torch.cuda.synchronize()
        rows = eng.profile_read()
        eng.profile(False)
        print(f"B={B} fcln={v}  " + "  ".join(f"{k}: {ms / 3:.3f} ms / {n // 3}" for k, (ms, n) in rows.items() if n), flush=True)
