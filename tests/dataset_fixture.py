"""A small on-disk dataset in the reference's formats (tests only): PCM16 WAVs under mix/ s1/ s2/ and
``np.savez_compressed(embedding=(512, Tv) f32)`` lip embeddings (make_embeddings.py:69), plus the index entries
BaseDataset.__getitem__ consumes (base_dataset.py:70-98)."""
import os
import wave

import numpy as np


def write_wav(path, x, sr=8000):
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())
    return pcm.astype(np.float32) / 32768.0


def make_dataset(root, n, T, Tv=50, emb=512, sr=8000, seed=0):
    rng = np.random.default_rng(seed)
    for d in ("mix", "s1", "s2", "emb"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    entries, truth = [], []
    for i in range(n):
        s1 = 0.1 * rng.standard_normal(T)
        s2 = 0.1 * rng.standard_normal(T)
        paths = {k: os.path.join(root, k, f"utt{i:04d}.wav") for k in ("mix", "s1", "s2")}
        q = {"s1": write_wav(paths["s1"], s1, sr), "s2": write_wav(paths["s2"], s2, sr)}
        q["mix"] = write_wav(paths["mix"], s1 + s2, sr)
        e = {}
        for k in ("s1", "s2"):
            e[k] = rng.standard_normal((emb, Tv)).astype(np.float32)
            np.savez_compressed(os.path.join(root, "emb", f"utt{i:04d}_{k}.npz"), embedding=e[k])
        entries.append({"mix_wav_path": paths["mix"], "s1_wav_path": paths["s1"], "s2_wav_path": paths["s2"],
                        "s1_video_path": None, "s2_video_path": None,
                        "s1_embedding_path": os.path.join(root, "emb", f"utt{i:04d}_s1.npz"),
                        "s2_embedding_path": os.path.join(root, "emb", f"utt{i:04d}_s2.npz")})
        truth.append({**q, "s1_embedding": e["s1"], "s2_embedding": e["s2"]})
    return entries, truth
