#!/usr/bin/env python3
"""GPU box: shader clock and socket power (rocm-smi, read only) while the B = 16 forward runs as usual (sub-batches on three
streams), serialised (option serialize: one kernel at a time) and while the chip idles -- the evidence behind "the chip is
power-limited under this workload" (DESIGN.md section 3.5 / 0).   python3 tools/power_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
eng = DptnEngine(DPTN_AV, dev)
eng.bind(params_to_device(synthetic_state_dict(DPTN_AV, 0), dev))
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(DPTN_AV, B=16, T=32000, Tv=50, seed=1).items()}
args = (t["mix"], t["s1_embedding"], t["s2_embedding"])


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:      # noqa: BLE001
        return None, None, repr(e)
    sclk = re.search(r"sclk clock level:?\s*\S*\s*\((\d+)Mhz\)", out)
    pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None), (float(pw.group(1)) if pw else None), out


def sample(label, seconds, work):
    stop, vals = threading.Event(), []

    def watcher():
        while not stop.is_set():
            s, p, _ = smi()
            vals.append((s, p))
            time.sleep(0.2)
    th = threading.Thread(target=watcher)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        if work:
            for _ in range(10):
                eng.forward(*args)
            torch.cuda.synchronize()
            n += 10
        else:
            time.sleep(0.2)
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    s = [v[0] for v in vals if v[0]]
    p = [v[1] for v in vals if v[1]]
    ms = f"{1e3 * dt / n:.2f} ms per forward" if n else "-"
    print(f"{label:34s} sclk MHz min/mean/max {min(s) if s else None}/{sum(s) / len(s) if s else 0:.0f}/{max(s) if s else None}  "
          f"power W mean {sum(p) / len(p) if p else 0:.0f} max {max(p) if p else None}  ({len(vals)} samples)  {ms}", flush=True)


_, _, raw = smi()
print(raw[:1500] if raw else "no rocm-smi output")
sample("idle", 3, False)
sample("forward, as run (3 streams)", 8, True)
eng.set_option("serialize", 1)
sample("forward, serialised", 8, True)
eng.set_option("serialize", 0)
eng.set_option("fuse_pre128", 0)
sample("forward, K4 + lstm16 (as run)", 8, True)
