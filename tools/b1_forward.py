"""One mixture at a time (the bs = 1 protocol), N synchronised forwards -- the program tools/gpu_b1_timeline.sh traces.
usage: b1_forward.py [dptn_av|dptn_audio] [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

cfg = DPTN_AUDIO if len(sys.argv) > 1 and sys.argv[1] == "dptn_audio" else DPTN_AV
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=1, T=32000, Tv=50, seed=1).items()}
for _ in range(calls):
    eng.forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    torch.cuda.synchronize()
print("done")
