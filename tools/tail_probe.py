"""Head / tail launches of the DPTN-AV forward (B = 16 x 4 s) with the sub-batch launches serialised: device time per class with
fcln.hip (option fcln = 1) and with the GEMM engine (0) for the last path's FFN + LayerNorm 2 and the separation conv.
python3 tools/tail_probe.py [dptn_av|dptn_audio]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "dptn_av"
cfg = {"dptn_av": DPTN_AV, "dptn_audio": DPTN_AUDIO}[name]
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=16, T=32000, Tv=50, seed=1).items()}
args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
eng.set_option("serialize", 1)
for rnd in range(2):
    for v in (0, 1):
        eng.set_option("fcln", v)
        for _ in range(3):
            eng.forward(*args)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(5):
            eng.forward(*args)
        torch.cuda.synchronize()
        rows = eng.profile_read()
        eng.profile(False)
        print(f"{name} fcln={v}  " + "  ".join(f"{k}: {ms / 5:.3f} ms / {n // 5}" for k, (ms, n) in rows.items() if n), flush=True)
