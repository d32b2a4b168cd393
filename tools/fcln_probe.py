"""fc + LayerNorm + residual of the DPRNN blocks (BASELINE configs[4] head, B = 32 x 128000 samples): time of the launches per
forward with the dedicated kernel (option fcln = 1 / 2) and with the GEMM engine (0), serialised launches (option
serialize), same process.   python3 tools/fcln_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPRNN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = DPRNN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=32, T=128000, Tv=50, seed=1).items()}
args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
eng.set_option("serialize", 1)
for rnd in range(2):
    for v in (0, 1, 2):
        eng.set_option("fcln", v)
        for _ in range(2):
            eng.forward(*args)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(3):
            eng.forward(*args)
        torch.cuda.synchronize()
        rows = eng.profile_read()
        eng.profile(False)
        print(f"fcln={v}  " + "  ".join(f"{k}: {ms / 3:.3f} ms / {n // 3}" for k, (ms, n) in rows.items() if n), flush=True)
