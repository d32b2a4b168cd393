"""Forward latency against the recurrence tile height (GPU box):  python tools/lstm4_latency.py [B ...]

The bs = 1 protocol of the reference's profiler.py (one 4 s mixture, synchronised wall time per forward) and small
batches, with the 4-sequence recurrence (lstm4.hip: option lstm4 = 0 off / 1 when it fits the chip in one round / 2
always)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16]
    dev = torch.device("cuda:0")
    for cfg_name, cfg in (("dptn_av", DPTN_AV), ("dptn_audio", DPTN_AUDIO)):
        eng = DptnEngine(cfg, dev)
        eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
        for B in batches:
            t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=1).items()}
            ref = None
            row = []
            for lstm4 in (0, 1, 2):
                eng.set_option("lstm4", lstm4)
                for _ in range(3):
                    out = eng.forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
                torch.cuda.synchronize()
                ts = []
                for _ in range(10):
                    t0 = time.perf_counter()
                    out = eng.forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
                    torch.cuda.synchronize()
                    ts.append(1e3 * (time.perf_counter() - t0))
                o = out[0].float().cpu().numpy()
                if ref is None:
                    ref = o
                db = 10 * np.log10((ref ** 2).sum() / max(((o - ref) ** 2).sum(), 1e-30))
                row.append(f"lstm4={lstm4}: {np.mean(ts):7.3f} ms (min {min(ts):7.3f}, {db:5.1f} dB vs lstm4=0)")
            print(f"{cfg_name} B={B:2d}  " + "  ".join(row), flush=True)
        del eng


if __name__ == "__main__":
    main()
