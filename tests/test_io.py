"""Host-side data path either side of the forward (formats restated from the reference, see speech_separation_amd/io.py)."""
import os

import numpy as np
import torch

from speech_separation_amd.io import PinnedBatcher, collate, load_object, save_predictions


def _items(tmp_path, n=3, T=64):
    rng = np.random.default_rng(0)
    items = []
    for i in range(n):
        e = rng.standard_normal((512, 50)).astype(np.float32)
        p = os.path.join(tmp_path, f"emb{i}.npz")
        np.savez_compressed(p, embedding=e)                       # make_embeddings.py:69
        items.append({"mix": torch.randn(1, T), "s1": torch.randn(1, T), "s2": torch.randn(1, T),
                      "s1_embedding": load_object(p), "s2_embedding": load_object(p), "s1_video": None,
                      "mix_spectrogram": torch.randn(1, 8, 5), "audio_path": f"/data/mix/utt{i}.wav"})
        assert items[-1]["s1_embedding"].shape == (1, 512, 50) and np.array_equal(items[-1]["s1_embedding"][0].numpy(), e)
    return items


def test_collate_semantics(tmp_path):
    items = _items(str(tmp_path))
    b = collate(items)
    assert b["mix"].shape == (3, 64) and b["s1_embedding"].shape == (3, 512, 50) and b["mix_spectrogram"].shape == (3, 8, 5)
    assert b["s1_video"] is None and b["audio_path"] == [f"/data/mix/utt{i}.wav" for i in range(3)]
    assert "s2_video" not in b
    assert torch.equal(b["s2"][1], items[1]["s2"][0])


def test_pinned_batcher_equals_collate_on_cpu(tmp_path):
    items = _items(str(tmp_path))
    want = collate(items)
    got = PinnedBatcher("cpu").to_device(items)
    assert set(got) == set(want)
    for k, v in want.items():
        assert (got[k] is None and v is None) or (isinstance(v, list) and got[k] == v) or torch.equal(got[k], v)


def test_prediction_files_have_the_reference_layout(tmp_path):
    items = _items(str(tmp_path))
    b = collate(items)
    b["s1_pred"], b["s2_pred"] = b["s1"] * 2, b["s2"] * 3
    paths = save_predictions(b, str(tmp_path / "saved" / "val"))
    assert [os.path.basename(p) for p in paths] == ["utt0.pth", "utt1.pth", "utt2.pth"]
    d = torch.load(paths[1])
    assert set(d) == {"s1_pred", "s2_pred", "s1_true", "s2_true"} and d["s1_pred"].shape == (64,)
    assert torch.equal(d["s2_pred"], b["s2"][1] * 3) and torch.equal(d["s1_true"], b["s1"][1])
    b["s1"] = b["s2"] = None                                       # no ground truth: only the predictions are written
    d = torch.load(save_predictions(b, str(tmp_path / "saved" / "test"))[0])
    assert set(d) == {"s1_pred", "s2_pred"}


def test_wav_and_item_loading(tmp_path):
    """load_audio == what torchaudio.load returns for PCM16 (first channel, / 32768, shape (1, T)); load_item builds the
    element BaseDataset.__getitem__ builds (base_dataset.py:56-135) for the keys this path consumes."""
    import pytest
    from dataset_fixture import make_dataset
    from speech_separation_amd.io import load_audio, load_item
    entries, truth = make_dataset(str(tmp_path), n=2, T=801, Tv=7, emb=16)
    a = load_audio(entries[1]["mix_wav_path"], target_sr=8000)
    assert a.shape == (1, 801) and a.dtype == torch.float32 and np.array_equal(a[0].numpy(), truth[1]["mix"])
    up = load_audio(entries[1]["mix_wav_path"], target_sr=16000)      # a rate mismatch is resampled (base_dataset.py:146-147)
    assert up.shape == (1, 1602) and up.dtype == torch.float32 and torch.isfinite(up).all()
    it = load_item(entries[0], target_sr=8000)
    assert it["audio_path"] == entries[0]["mix_wav_path"] and it["s1_video"] is None
    assert it["s1_embedding"].shape == (1, 16, 7) and np.array_equal(it["s2_embedding"][0].numpy(), truth[0]["s2_embedding"])
    assert np.array_equal(it["s2"][0].numpy(), truth[0]["s2"])
    no_gt = load_item({"mix_wav_path": entries[0]["mix_wav_path"], "s1_wav_path": None, "s1_embedding_path": None})
    assert no_gt["s1"] is None and no_gt["s1_embedding"] is None
    b = collate([load_item(e) for e in entries])
    assert b["mix"].shape == (2, 801) and b["s1_video"] is None


def test_resample_properties():
    """io.resample restates torchaudio.functional.resample's defaults (third-party, not installed: parity unpinned), so it is
    checked through what a band-limited rate conversion must do: length ceil(T new / orig), identity for equal rates, a
    tone well inside both bands keeps frequency, amplitude and phase (away from the edges), a tone above the new Nyquist is
    removed, batch rows are independent, and down- then up-sampling a band-limited signal gives it back."""
    import math
    from speech_separation_amd.io import resample
    sr = 16000
    t = torch.arange(16000, dtype=torch.float64) / sr
    tone = torch.sin(2 * math.pi * 440.0 * t).float()[None]
    assert resample(tone, sr, sr) is tone
    for new in (8000, 22050, 48000, 11025):
        y = resample(tone, sr, new)
        n = int(math.ceil(16000 * new / sr))
        assert y.shape == (1, n)
        tn = torch.arange(n, dtype=torch.float64) / new
        want = torch.sin(2 * math.pi * 440.0 * tn).float()
        core = slice(200, n - 200)
        err = (y[0, core] - want[core]).pow(2).mean().sqrt()
        assert float(err) < 5e-3, (new, float(err))
    # 6 kHz is above the 4 kHz Nyquist of 8 kHz: filtered out (stop band of a 6-zero-crossing Hann-windowed sinc: < -30 dB)
    high = torch.sin(2 * math.pi * 6000.0 * t).float()[None]
    assert float(resample(high, sr, 8000)[0, 200:-200].abs().max()) < 0.05
    both = torch.cat([tone, high])
    yb = resample(both, sr, 8000)
    assert torch.allclose(yb[0:1], resample(tone, sr, 8000), atol=1e-6) and torch.allclose(yb[1:2], resample(high, sr, 8000), atol=1e-6)
    back = resample(resample(tone, sr, 8000), 8000, sr)
    assert back.shape == tone.shape and float((back - tone)[0, 400:-400].abs().max()) < 1e-2
