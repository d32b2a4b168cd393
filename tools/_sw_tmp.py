import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speech_separation_amd.engine import DptnEngine, params_to_device
from speech_separation_amd.spec import DPTN_AV, DPTNConfig, synthetic_inputs, synthetic_state_dict
dev = torch.device("cuda:0")
cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2})
eng = DptnEngine(cfg, dev); eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
g = eng.capture_forward(1, 8000, 50)
keep = []
for s in (3, 4, 5, 6):
    inp = synthetic_inputs(cfg, B=1, T=8000, Tv=50, seed=s)
    print("PYHOST seed", s, {k: hex(v.ctypes.data) for k, v in inp.items()}, file=sys.stderr, flush=True)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    del inp
    keep.append(t)
    print("PYMARK copy+replay seed", s, file=sys.stderr, flush=True)
    r = g(t["mix"], t["s1_embedding"], t["s2_embedding"])
    torch.cuda.synchronize()
    print("PYMARK replay done seed", s, file=sys.stderr, flush=True)
    c = [x.clone() for x in r]; keep.append(c); torch.cuda.synchronize()
print("ok")
