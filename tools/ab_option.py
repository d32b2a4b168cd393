"""Same-box A/B of an engine option (boxes differ by 1-2 %, so two builds / settings are only comparable inside one process):
forward at B = 16 and bs = 1, DPTN-AV and DPTN audio-only, alternating the option's values.
usage: ab_option.py <option> <value_a> <value_b> [rounds] [configs]      configs: comma-separated among dptn_av, dptn_audio,
dprnn_av (B = 32 x 128000 samples, BASELINE configs[4]); default dptn_av,dptn_audio"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
if os.environ.get("AB_LIB"):          # A/B of two BUILDS: AB_LIB=<path to the other .so>, run the script once per build
    import speech_separation_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPRNN_AV, DPTN_AUDIO, DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
ALL = {"dptn_av": (DPTN_AV, 32000, ((16, 20), (1, 40))), "dptn_audio": (DPTN_AUDIO, 32000, ((16, 20), (1, 40))),
       "dprnn_av": (DPRNN_AV, 128000, ((32, 3),))}
names = sys.argv[5].split(",") if len(sys.argv) > 5 else ["dptn_av", "dptn_audio"]
for name in names:
    cfg, T, sizes = ALL[name]
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
    for B, reps in sizes:
        t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=1).items()}
        args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
        res = {va: [], vb: []}
        for r in range(rounds):
            for v in (va, vb):
                eng.set_option(opt, v)
                for _ in range(3):
                    eng.forward(*args)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.forward(*args)
                torch.cuda.synchronize()
                res[v].append(1e3 * (time.perf_counter() - t0) / reps)
        print(f"{name} B={B:2d}: {opt}={va}: {np.mean(res[va]):.3f} ms ({' '.join(f'{x:.3f}' for x in res[va])})   "
              f"{opt}={vb}: {np.mean(res[vb]):.3f} ms ({' '.join(f'{x:.3f}' for x in res[vb])})", flush=True)
    del eng
    torch.cuda.empty_cache()
