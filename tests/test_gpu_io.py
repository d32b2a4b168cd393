"""The input side and the prediction writer on the GPU: pinned-memory batching with asynchronous H2D
must hand the forward exactly what collate() builds, batch after batch, and the whole load -> H2D -> forward -> metric ->
D2H -> write pipeline must produce the files and the metric of the plain synchronous loop (inferencer.py:98-147)."""
import os

import numpy as np
import pytest
import torch

from dataset_fixture import make_dataset
from speech_separation_amd.spec import synthetic_state_dict

pytestmark = pytest.mark.gpu


def _model(dev, blocks=1):
    from speech_separation_amd import DPTNAVWavEncDec
    model = DPTNAVWavEncDec(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128,
                            num_blocks=blocks, chunk_size=150, step_size=75, num_heads=4)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
    return model.to(dev).eval()


def test_pinned_batcher_on_the_gpu_equals_collate(tmp_path):
    """Several DISTINCT batches back to back with a forward in between: every device tensor must equal collate()'s, also
    the ones whose pinned staging buffer and device memory are recycled while an earlier forward is still running."""
    from speech_separation_amd.io import PinnedBatcher, collate, load_item
    dev = torch.device("cuda:0")
    entries, _ = make_dataset(str(tmp_path), n=24, T=4000)
    items = [load_item(e, 8000) for e in entries]
    model = _model(dev)
    # overlapped run: nothing synchronises inside the loop, the input batch is dropped as soon as its forward is enqueued
    # (so the allocator may hand its memory to the next batch's copies), only the outputs are kept
    batcher = PinnedBatcher(dev)
    outs = []
    with torch.no_grad():
        for b in range(6):
            batch = batcher.to_device(items[4 * b:4 * b + 4])
            o = model(**batch)
            outs.append((o["s1_pred"], o["s2_pred"]))
            del batch, o
        torch.cuda.synchronize()
    # synchronous replay: tensors equal collate()'s, and the overlapped run produced the same outputs
    batcher = PinnedBatcher(dev)
    with torch.no_grad():
        for b in range(6):
            chunk = items[4 * b:4 * b + 4]
            want = collate(chunk)
            got = batcher.to_device(chunk)
            out = model(**got)
            ref = model(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in want.items()})
            assert got["audio_path"] == want["audio_path"] and got["s1_video"] is None
            for k in ("mix", "s1", "s2", "s1_embedding", "s2_embedding"):
                assert got[k].device == dev and torch.equal(got[k].cpu(), want[k]), (b, k)
            assert torch.equal(out["s1_pred"], ref["s1_pred"]) and torch.equal(out["s2_pred"], ref["s2_pred"])
            assert torch.equal(outs[b][0], ref["s1_pred"]) and torch.equal(outs[b][1], ref["s2_pred"]), b


def test_inference_pipeline_writes_the_same_files_as_the_synchronous_loop(tmp_path):
    from speech_separation_amd.evaluate import evaluate, run_inference
    from speech_separation_amd.io import collate, load_item, save_predictions
    from speech_separation_amd.metrics import SISNRiMetric
    dev = torch.device("cuda:0")
    n, bs = 37, 4                                             # 10 batches, the last one ragged
    entries, _ = make_dataset(str(tmp_path / "data"), n=n, T=4000)
    model = _model(dev)
    logs, stats = run_inference(model, entries, bs, [SISNRiMetric(name="SISNRiMetric")], save_dir=str(tmp_path / "fast"),
                                device=dev, workers=4, target_sr=8000)
    assert stats["items"] == n and stats["files"] == n and stats["items_per_s"] > 0
    # the plain loop: load, collate, .to(device), forward, metric, save_predictions -- one batch at a time
    batches = [collate([load_item(e, 8000) for e in entries[i:i + bs]]) for i in range(0, n, bs)]
    want = evaluate(model, batches, [SISNRiMetric(name="SISNRiMetric")])
    assert abs(logs["SISNRiMetric"] - want["SISNRiMetric"]) < 1e-4
    with torch.no_grad():
        for b in batches:
            db = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
            db.update(model(**db))
            save_predictions(db, str(tmp_path / "slow"))
    names = sorted(os.listdir(tmp_path / "slow"))
    assert names == sorted(os.listdir(tmp_path / "fast")) and len(names) == n
    for f in names:
        a, b = torch.load(tmp_path / "fast" / f), torch.load(tmp_path / "slow" / f)
        assert set(a) == set(b) == {"s1_pred", "s2_pred", "s1_true", "s2_true"}
        assert all(a[k].dim() == 1 and torch.equal(a[k], b[k]) for k in a), f
    print(f"pipeline: {stats['items_per_s']:.1f} items/s over {n} items")
