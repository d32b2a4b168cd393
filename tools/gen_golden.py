#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference model on CPU.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference -- it sees the small .npz fixtures this script writes.
The reference package __init__ pulls librosa/cv2 (absent here), so the model
files are imported through stub packages (SURVEY.md section 8c recipe).

Weights and inputs are NOT stored for the large cases: both sides regenerate
them from speech_separation_amd.spec.synthetic_state_dict / synthetic_inputs
(numpy PCG64 -> identical on every machine); a sha256 of the weight bytes is
stored so a drift is detected instead of silently mis-compared.

Usage:  python tools/gen_golden.py [name ...]  (rewrites tests/golden/, or only the named fixtures)
"""
from __future__ import annotations

import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.spec import (DPRNN_AUDIO, DPTN_AV, DPTN_AUDIO, DPTN_TINY, DPTNConfig,  # noqa: E402
                                        state_dict_spec, synthetic_inputs, synthetic_state_dict)

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name, path in [("src", f"{REF}/src"), ("src.model", f"{REF}/src/model"), ("src.loss", f"{REF}/src/loss")]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    sys.path.insert(0, REF)
    dptn_wav = importlib.import_module("src.model.dptn_wav")
    dptn_wav.DPRNNEncDec = importlib.import_module("src.model.dprnn").DPRNNEncDec
    losses = importlib.import_module("src.loss.ss_losses")
    return dptn_wav, losses


def weights_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


def build_reference(dptn_wav, cfg: DPTNConfig, sd):
    kw = dict(num_features=cfg.num_features, kernel_size_enc=cfg.kernel_size_enc, hidden_dim=cfg.hidden_dim,
              num_blocks=cfg.num_blocks, chunk_size=cfg.chunk_size, step_size=cfg.step_size,
              num_heads=cfg.num_heads, dropout=cfg.dropout, bidir=cfg.bidir)
    if cfg.arch == "dprnn":
        assert cfg.audio_only, "the reference only defines the audio-only DPRNNEncDec"
        kw.pop("num_heads"), kw.pop("dropout")
        model = dptn_wav.DPRNNEncDec(**kw)
    elif cfg.audio_only:
        model = dptn_wav.DPTNWavEncDec(**kw)
    else:
        model = dptn_wav.DPTNAVWavEncDec(video_emb_size=cfg.video_emb_size, hidden_video=cfg.hidden_video, **kw)
    ref_keys = list(model.state_dict().keys())
    our_keys = [k for k, _ in state_dict_spec(cfg)]
    assert ref_keys == our_keys, "state_dict key order drifted from spec"
    for k, s in state_dict_spec(cfg):
        assert tuple(model.state_dict()[k].shape) == s, (k, s)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return model.eval()


def run_with_taps(model, cfg: DPTNConfig, inp):
    """Forward with hooks -> dict of stage tensors (reference layouts)."""
    taps = {}
    hooks = []

    def tap(name, mod, which="out"):
        def fn(_m, args, out):
            t = args[0] if which == "in" else out
            if isinstance(t, (list, tuple)):
                t = torch.stack(list(t), 0)
            taps[name] = t.detach().clone().numpy()
        hooks.append(mod.register_forward_hook(fn))

    tap("enc_conv", model.encoder)                    # A2 (before video fusion)
    tap("encoded", model.dprnn, "in")                 # A3 fused latent (B,N,L)
    tap("chunked", model.dprnn.segmenter)             # A4 (B,N,S,K)
    for b, blk in enumerate(model.dprnn.model):
        tap(f"blk{b}_intra", blk.intra_chunk_block)   # (B*S,K,N)
        tap(f"blk{b}_inter", blk.inter_chunk_block)   # (B*K,S,N)
        tap(f"blk{b}_out", blk)                       # (B,N,S,K)
    tap("sep", model.dprnn.speakers_separation)       # (B,2N,S,K)
    tap("ola", model.dprnn.overladd)                  # (B,2N,ola)
    tap("masks", model.dprnn)                         # stacked (2,B,N,L)
    with torch.no_grad():
        batch = {k: torch.from_numpy(v) for k, v in inp.items() if k in ("mix", "s1_embedding", "s2_embedding")}
        # extra key must be swallowed by **batch exactly like trainer.py:40 passes it
        out = model(mix_spectrogram=torch.zeros(1), **batch)
    for h in hooks:
        h.remove()
    taps["s1_pred"] = out["s1_pred"].numpy()
    taps["s2_pred"] = out["s2_pred"].numpy()
    return taps


def loss_and_metric(losses, taps, inp):
    """A10 (reference code) and A11 (restated: torchmetrics is not installed)."""
    t = {k: torch.from_numpy(v) for k, v in {**inp, **{k: taps[k] for k in ("s1_pred", "s2_pred")}}.items()}
    crit = losses.SiSNRWavLoss()
    loss = crit(s1_pred=t["s1_pred"], s2_pred=t["s2_pred"], s1=t["s1"], s2=t["s2"])["loss"].item()
    single = losses.SiSNRLoss()
    pair = {f"sisnr_loss_{a}_{b}": single(t[f"{a}_pred"], t[b]).item() for a in ("s1", "s2") for b in ("s1", "s2")}
    return {"pit_loss": np.float64(loss), **{k: np.float64(v) for k, v in pair.items()}}


def subsample(a: np.ndarray, step: int) -> np.ndarray:
    return np.ascontiguousarray(a.reshape(-1)[::step])


GRAD_STRIDE = 37      # big gradients are stored as every 37th element (coprime with every tensor width on the path)
GRAD_FULL_BELOW = 4096


def reference_gradients(dptn_wav, losses, cfg: DPTNConfig, sd, inp):
    """The reference's OWN training step up to loss.backward() (trainer.py:38-47): model.train(), outputs = model(**batch),
    SiSNRWavLoss (ss_losses.py:21-26,96-130), loss.backward() -- run twice: in fp32 as the trainer runs it, and with the
    same modules cast to fp64 (`model.double()`, the reference's code, exact to ~1e-15).  The fp64 run's gradients are
    the fixture's truth; the fp32 run's agreement with them, per parameter (`ref32db.*`), is the REFERENCE'S OWN rounding
    noise (bias / LayerNorm gradients are sums over 10^5..10^6 tokens: 50-60 dB in fp32).  dropout must be 0.0 for a
    deterministic fixture (train-mode attention dropout is a torch RNG stream); nn.LSTM(dropout=1) is a
    no-op for one layer (dptn.py:23-29)."""
    assert cfg.dropout == 0.0
    from oracle import dptn_oracle as O

    def run(dtype):
        model = build_reference(dptn_wav, cfg, sd).train().to(dtype)
        batch = {k: torch.from_numpy(v).to(dtype) for k, v in inp.items()}
        model.zero_grad()
        with torch.enable_grad():
            batch.update(model(mix_spectrogram=torch.zeros(1), **batch))
            loss = losses.SiSNRWavLoss()(**batch)["loss"]
            loss.backward()
        return (float(loss.item()), {k: p.grad.detach().double().numpy() for k, p in model.named_parameters()},
                {k: batch[k].detach().numpy() for k in ("s1_pred", "s2_pred")})

    loss32, g32, pred32 = run(torch.float32)
    loss64, g64, _ = run(torch.float64)
    out = {"val.loss": np.float64(loss32), "val.loss64": np.float64(loss64), "tap.s1_pred": pred32["s1_pred"],
           "tap.s2_pred": pred32["s2_pred"]}
    for k, g in g64.items():
        out[f"norm.{k}"] = np.float64(np.sqrt((g ** 2).sum()))
        out[f"ref32db.{k}"] = np.float64(O.agreement_db(g32[k], g))
        out[f"grad.{k}"] = (g.copy() if g.size <= GRAD_FULL_BELOW else subsample(g, GRAD_STRIDE)).astype(np.float32)
    # what clip_grad_norm_ sees (base_trainer.py:383-391), exact and as the reference's fp32 step computes it
    out["val.grad_norm"] = np.float64(np.sqrt(sum(float((g ** 2).sum()) for g in g64.values())))
    out["val.grad_norm32"] = np.float64(np.sqrt(sum(float((g ** 2).sum()) for g in g32.values())))
    return out


def gradient_fixtures(dptn_wav, losses, only):
    """config 4's gradients pinned to the reference itself (not to autograd on this repo's port)."""
    cases = [
        ("grad_tiny_av", DPTNConfig(**{**DPTN_TINY.to_dict(), "dropout": 0.0}), dict(B=2, T=209, Tv=9), 7, 11),
        ("grad_mid_av", DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2, "dropout": 0.0}), dict(B=2, T=8000, Tv=50), 0, 123),
        ("grad_full_av", DPTNConfig(**{**DPTN_AV.to_dict(), "dropout": 0.0}), dict(B=1, T=32000, Tv=50), 0, 123),
        # the reference's audio-only DPTNWavEncDec (model/dptn_wav.yaml: 64 features), trainable here since round 3
        ("grad_mid_audio", DPTNConfig(**{**DPTN_AUDIO.to_dict(), "num_blocks": 2, "dropout": 0.0}), dict(B=2, T=8000, Tv=50), 0, 123),
        # the reference's DPRNNEncDec (model/dprnn.yaml: 64 features, kernel 2, chunks of 250)
        ("grad_mid_dprnn", DPTNConfig(**{**DPRNN_AUDIO.to_dict(), "num_blocks": 2, "dropout": 0.0}), dict(B=2, T=3000, Tv=50), 0, 123),
    ]
    for name, cfg, shp, wseed, iseed in cases:
        if only and name not in only:
            continue
        sd = synthetic_state_dict(cfg, seed=wseed)
        inp = synthetic_inputs(cfg, seed=iseed, **shp)
        rec = reference_gradients(dptn_wav, losses, cfg, sd, inp)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), cfg=np.array(repr(cfg.to_dict())),
                            shape=np.array([shp["B"], shp["T"], shp["Tv"]]), seeds=np.array([wseed, iseed]),
                            stride=np.array(GRAD_STRIDE), full_below=np.array(GRAD_FULL_BELOW),
                            digest=np.array(weights_digest(sd)), **rec)
        print(name, "loss", rec["val.loss"], rec["val.loss64"], "grad norm", rec["val.grad_norm"], rec["val.grad_norm32"],
              "reference fp32 vs fp64, worst parameter:", min((float(v), k) for k, v in rec.items() if k.startswith("ref32db.")))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    dptn_wav, losses = import_reference()

    # ---- 1. tiny config: every stage tensor, weights and inputs stored ----
    for name, cfg in (("tiny_av", DPTN_TINY), ("tiny_audio", DPTNConfig(**{**DPTN_TINY.to_dict(), "audio_only": True})),
                      ("tiny_unidir", DPTNConfig(**{**DPTN_TINY.to_dict(), "bidir": False})),
                      ("tiny_dprnn", DPTNConfig(**{**DPTN_TINY.to_dict(), "audio_only": True, "arch": "dprnn",
                                                   "kernel_size_enc": 2})),
                      ("tiny_dprnn_unidir", DPTNConfig(**{**DPTN_TINY.to_dict(), "audio_only": True, "arch": "dprnn",
                                                          "bidir": False}))):
        if set(sys.argv[1:]) and name not in set(sys.argv[1:]):
            continue
        sd = synthetic_state_dict(cfg, seed=7)
        inp = synthetic_inputs(cfg, B=2, T=209, Tv=9, seed=11)
        model = build_reference(dptn_wav, cfg, sd)
        taps = run_with_taps(model, cfg, inp)
        extra = loss_and_metric(losses, taps, inp)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), cfg=np.array(repr(cfg.to_dict())),
                            digest=np.array(weights_digest(sd)),
                            **{f"w.{k}": v for k, v in sd.items()}, **{f"in.{k}": v for k, v in inp.items()},
                            **{f"tap.{k}": v for k, v in taps.items()}, **{f"val.{k}": v for k, v in extra.items()})
        print(name, "stages:", len(taps), "loss", extra["pit_loss"])

    # ---- 2. real feature sizes, short audio, 2 blocks: outputs + strided taps ----
    cases = [
        ("mid_av", DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2}), dict(B=2, T=8000, Tv=50)),
        ("mid_audio", DPTNConfig(**{**DPTN_AUDIO.to_dict(), "num_blocks": 2}), dict(B=2, T=8000, Tv=50)),
        ("full_av", DPTN_AV, dict(B=1, T=32000, Tv=50)),
        ("mid_dprnn", DPTNConfig(**{**DPRNN_AUDIO.to_dict(), "num_blocks": 2}), dict(B=2, T=3000, Tv=50)),
        # BASELINE configs[1] and configs[4] at their real sizes (one mixture each; the batch sizes of the configs are
        # covered on the GPU by batch-independence against these rows): model/dptn_wav.yaml, model/dprnn.yaml
        ("full_audio", DPTN_AUDIO, dict(B=1, T=32000, Tv=50)),
        ("full_dprnn", DPRNN_AUDIO, dict(B=1, T=128000, Tv=50)),
    ]
    only = set(sys.argv[1:])       # `python tools/gen_golden.py full_audio full_dprnn` regenerates just those
    for name, cfg, shp in cases:
        if only and name not in only:
            continue
        sd = synthetic_state_dict(cfg, seed=0)
        inp = synthetic_inputs(cfg, seed=123, **shp)
        model = build_reference(dptn_wav, cfg, sd)
        taps = run_with_taps(model, cfg, inp)
        extra = loss_and_metric(losses, taps, inp)
        keep = {"s1_pred": taps["s1_pred"], "s2_pred": taps["s2_pred"]}
        big = name in ("full_audio", "full_dprnn")     # 16 M-element stage tensors: a few of them, thinly sampled
        for k, v in taps.items():
            if k in keep:
                continue
            if not big:
                keep[f"strided97.{k}"] = subsample(v, 97)
            elif k in ("encoded", "blk0_intra", f"blk{cfg.num_blocks - 1}_out", "sep", "masks"):
                keep[f"strided9973.{k}"] = subsample(v, 9973)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), cfg=np.array(repr(cfg.to_dict())),
                            shape=np.array([shp["B"], shp["T"], shp["Tv"]]), digest=np.array(weights_digest(sd)),
                            **{f"tap.{k}": v for k, v in keep.items()}, **{f"val.{k}": v for k, v in extra.items()})
        print(name, "loss", extra["pit_loss"], "rms", float(np.sqrt((taps['s1_pred'] ** 2).mean())))


    # ---- 2b. the reference's loss.backward() (config 4): loss, every parameter's gradient norm, gradient samples ----
    gradient_fixtures(dptn_wav, losses, only)

    # ---- 3. Conv-TasNet (BASELINE configs[0], CPU-only reference case): outputs for seeded weights ----
    if only and "convtasnet" not in only:
        return
    from oracle.convtasnet_stock import convtasnet_spec, synthetic_convtasnet_weights
    ct = importlib.import_module("src.model.convtasnet").ConvTasNet().eval()
    assert [(k, tuple(v.shape)) for k, v in ct.state_dict().items()] == convtasnet_spec(), "ConvTasNet spec drifted"
    wsd = synthetic_convtasnet_weights(seed=0)
    ct.load_state_dict({k: torch.from_numpy(v) for k, v in wsd.items()}, strict=True)
    mixc = synthetic_inputs(DPTN_AUDIO, B=2, T=4000, seed=21)["mix"]
    with torch.no_grad():
        outc = ct(mix=torch.from_numpy(mixc))
    np.savez_compressed(os.path.join(OUT, "convtasnet.npz"), digest=np.array(weights_digest(wsd)),
                        s1_pred=outc["s1_pred"].numpy(), s2_pred=outc["s2_pred"].numpy())
    print("convtasnet", outc["s1_pred"].shape, float(outc["s1_pred"].abs().mean()))


if __name__ == "__main__":
    main()
