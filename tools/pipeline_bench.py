#!/usr/bin/env python3
"""GPU box: items/s of the whole inference pipeline -- WAV + .npz loading on loader threads, pinned
batching + async H2D, the DPTN-AV forward (B=16, T=32000), SI-SNRi, async D2H + one .pth per item -- next to the forward
alone.   python3 tools/pipeline_bench.py [items] [workers]"""
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dataset_fixture import make_dataset  # noqa: E402
from speech_separation_amd import DPTNAVWavEncDec  # noqa: E402
from speech_separation_amd.evaluate import run_inference  # noqa: E402
from speech_separation_amd.metrics import SISNRiMetric  # noqa: E402
from speech_separation_amd.spec import synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="dptnav_pipe_")
try:
    t0 = time.perf_counter()
    entries, _ = make_dataset(os.path.join(root, "data"), n=n, T=32000)
    print(f"dataset of {n} items written in {time.perf_counter() - t0:.1f} s")
    model = DPTNAVWavEncDec(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128,
                            num_blocks=6, chunk_size=150, step_size=75, num_heads=4)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
    model = model.to(dev).eval()
    met = [SISNRiMetric(name="SISNRiMetric")]
    run_inference(model, entries[:32], 16, met, save_dir=os.path.join(root, "warm"), device=dev, workers=workers, target_sr=8000)
    for label, save in (("load + H2D + forward + metric", None), ("... + D2H + one .pth per item", os.path.join(root, "out"))):
        logs, st = run_inference(model, entries, 16, met, save_dir=save, device=dev, workers=workers, target_sr=8000)
        print(f"{label:40s}: {st['items_per_s']:8.1f} items/s over {st['items']} items ({st['seconds']:.2f} s, {workers} loader threads)")
finally:
    shutil.rmtree(root, ignore_errors=True)
