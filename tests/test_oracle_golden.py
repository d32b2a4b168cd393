"""Pins oracle/dptn_oracle.py against tensors captured from the reference itself
(tools/gen_golden.py ran the reference on CPU; fixtures in tests/golden/)."""
import numpy as np
import pytest

from oracle import dptn_oracle as O
from speech_separation_amd.spec import synthetic_inputs, synthetic_state_dict
from tools.gen_golden import weights_digest

STAGES = ["enc_conv", "encoded", "chunked", "blk0_intra", "blk0_inter", "blk0_out", "blk1_intra", "blk1_inter",
          "blk1_out", "sep", "ola", "masks", "s1_pred", "s2_pred"]


def _split(z):
    w = {k[2:]: v for k, v in z.items() if k.startswith("w.")}
    i = {k[3:]: v for k, v in z.items() if k.startswith("in.")}
    t = {k[4:]: v for k, v in z.items() if k.startswith("tap.")}
    return w, i, t


@pytest.mark.parametrize("name", ["tiny_av", "tiny_audio", "tiny_unidir", "tiny_dprnn", "tiny_dprnn_unidir"])
def test_every_stage_matches_reference(golden, name):
    cfg, z = golden(name)
    w, inp, ref = _split(z)
    taps = {}
    out = O.forward(cfg, w, dtype=np.float32, taps=taps, **inp)
    taps.update(out)
    for st in STAGES:
        got = taps[st]
        if st == "masks":
            pass  # already (2,B,N,L)
        want = ref[st]
        assert got.shape == want.shape, st
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-12)
        assert err < 2e-5, (st, err)
    assert O.agreement_db(out["s1_pred"], ref["s1_pred"]) > 90
    assert O.agreement_db(out["s2_pred"], ref["s2_pred"]) > 90


@pytest.mark.parametrize("name", ["tiny_av", "tiny_audio"])
def test_loss_matches_reference(golden, name):
    cfg, z = golden(name)
    w, inp, ref = _split(z)
    for a in ("s1", "s2"):
        for b in ("s1", "s2"):
            got = O.si_snr_loss(ref[f"{a}_pred"].astype(np.float64), inp[b].astype(np.float64))
            assert abs(got - float(z[f"val.sisnr_loss_{a}_{b}"])) < 1e-3
    got = O.pit_loss(ref["s1_pred"], ref["s2_pred"], inp["s1"], inp["s2"])
    assert abs(got - float(z["val.pit_loss"])) < 1e-3


@pytest.mark.parametrize("name", ["tiny_av", "tiny_audio", "tiny_dprnn"])
def test_torch_loss_restatement_matches_reference(golden, name):
    """oracle/torch_stock.SiSNRWavLossTorch (the autograd-able oracle of the device loss kernel) reproduces the values the
    reference's own SiSNRWavLoss / SiSNRLoss gave on the reference's outputs."""
    import torch
    from oracle.torch_stock import SiSNRWavLossTorch
    cfg, z = golden(name)
    w, inp, ref = _split(z)
    t = {k: torch.from_numpy(v) for k, v in {**inp, "s1_pred": ref["s1_pred"], "s2_pred": ref["s2_pred"]}.items()}
    crit = SiSNRWavLossTorch()
    assert abs(float(crit(**t)["loss"]) - float(z["val.pit_loss"])) < 1e-4 * max(1.0, abs(float(z["val.pit_loss"])))
    for a in ("s1", "s2"):
        for b in ("s1", "s2"):
            assert abs(float(crit.pair(t[f"{a}_pred"], t[b])) - float(z[f"val.sisnr_loss_{a}_{b}"])) < 1e-4 * max(
                1.0, abs(float(z[f"val.sisnr_loss_{a}_{b}"])))


def test_metric_consistent_with_reference_loss(golden):
    """torchmetrics is unavailable (parity unpinned): SI-SNR dB must equal -loss/2 of the
    reference's own SiSNRLoss on the same pair (ss_losses.py:100-114)."""
    cfg, z = golden("tiny_av")
    w, inp, ref = _split(z)
    for a in ("s1", "s2"):
        for b in ("s1", "s2"):
            db = O.si_snr_db(ref[f"{a}_pred"], inp[b])
            assert abs(db - (-float(z[f"val.sisnr_loss_{a}_{b}"]) / 2)) < 1e-3


@pytest.mark.parametrize("name", ["mid_av", "mid_audio", "mid_dprnn"])
def test_real_feature_sizes_match_reference(golden, name):
    cfg, z = golden(name)
    B, T, Tv = (int(v) for v in z["shape"])
    sd = synthetic_state_dict(cfg, seed=0)
    assert weights_digest(sd) == str(z["digest"])
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=123)
    taps = {}
    out = O.forward(cfg, sd, dtype=np.float32, taps=taps, **inp)
    for k in ("s1_pred", "s2_pred"):
        assert O.agreement_db(out[k], z["tap." + k]) > 80, k
    for k in ("encoded", "blk0_intra", "blk1_out", "sep", "masks"):
        got = taps[k].reshape(-1)[::97]
        assert O.agreement_db(got, z["tap.strided97." + k]) > 80, k
    # metric delta on identical targets stays far below the 1e-3 dB budget
    d = abs(O.si_snri_metric(out["s1_pred"], out["s2_pred"], inp["s1"], inp["s2"], inp["mix"])
            - O.si_snri_metric(z["tap.s1_pred"], z["tap.s2_pred"], inp["s1"], inp["s2"], inp["mix"]))
    assert d < 1e-3


@pytest.mark.parametrize("name", ["tiny_av", "tiny_audio", "tiny_unidir", "tiny_dprnn", "tiny_dprnn_unidir"])
def test_stock_torch_composition_matches_reference(golden, name):
    """oracle/torch_stock.py (the cpu_baseline 'port') reproduces the reference outputs."""
    import torch
    from oracle.torch_stock import StockDPTN
    cfg, z = golden(name)
    w, inp, ref = _split(z)
    model = StockDPTN(cfg, w)
    out = model(**{k: torch.from_numpy(v) for k, v in inp.items()})
    for k in ("s1_pred", "s2_pred"):
        assert O.agreement_db(out[k].numpy(), ref[k]) > 100, k


def test_stock_torch_composition_real_sizes(golden):
    import torch
    from oracle.torch_stock import StockDPTN
    cfg, z = golden("mid_av")
    B, T, Tv = (int(v) for v in z["shape"])
    sd = synthetic_state_dict(cfg, seed=0)
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=123)
    out = StockDPTN(cfg, sd)(**{k: torch.from_numpy(v) for k, v in inp.items()})
    for k in ("s1_pred", "s2_pred"):
        assert O.agreement_db(out[k].numpy(), z["tap." + k]) > 100, k


def test_convtasnet_stock_composition_matches_reference():
    """BASELINE configs[0] (the reference's CPU-only case): oracle/convtasnet_stock.py vs the reference's ConvTasNet."""
    import os
    import torch
    from oracle import convtasnet_stock as CT
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "convtasnet.npz"))
    sd = CT.synthetic_convtasnet_weights(seed=0)
    assert weights_digest(sd) == str(z["digest"])
    assert sum(int(np.prod(s)) for _, s in CT.convtasnet_spec()) == 5_066_929
    from speech_separation_amd.spec import DPTN_AUDIO
    mix = synthetic_inputs(DPTN_AUDIO, B=2, T=4000, seed=21)["mix"]
    out = CT.forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(mix))
    for k in ("s1_pred", "s2_pred"):
        assert out[k].shape == (2, 4000) and O.agreement_db(out[k].numpy(), z[k]) > 100


def reference_gradient_report(z, grads, floor_db=60.0, margin_db=3.0):
    """Compare {state_dict key: gradient array} with a tests/golden/grad_*.npz record of the REFERENCE's loss.backward()
    (tools/gen_golden.py::reference_gradients; truth = the reference run in fp64, small tensors stored whole, big ones as
    every `stride`-th element).  A parameter passes when it agrees with the truth to `floor_db`, or to within `margin_db`
    of what the reference's OWN fp32 run achieves for it (`ref32db.*`: sums over 10^5..10^6 tokens are 50-60 dB in fp32).
    -> (failures [(key, dB, required dB)], worst dB, worst relative norm error)."""
    stride, full_below = int(z["stride"]), int(z["full_below"])
    keys = [k[5:] for k in z if k.startswith("grad.")]
    assert sorted(keys) == sorted(grads), set(keys) ^ set(grads)
    fails, worst_db, worst_norm = [], (1e9, ""), (-1.0, "")
    for k in keys:
        g = np.asarray(grads[k], dtype=np.float64)
        want = z["grad." + k]
        got = g.reshape(want.shape) if g.size <= full_below else g.reshape(-1)[::stride]
        db = O.agreement_db(got, want)
        need = min(floor_db, float(z["ref32db." + k]) - margin_db)
        if db < need:
            fails.append((k, round(db, 1), round(need, 1)))
        worst_db = min(worst_db, (db, k))
        nrm = float(np.sqrt((g ** 2).sum()))
        worst_norm = max(worst_norm, (abs(nrm - float(z["norm." + k])) / max(float(z["norm." + k]), 1e-30), k))
    return fails, worst_db, worst_norm


@pytest.mark.parametrize("name", ["grad_tiny_av", "grad_mid_av", "grad_mid_audio", "grad_mid_dprnn"])
def test_stock_autograd_matches_reference_gradients(golden, name):
    """The gradient oracle of the GPU backward tests (torch.autograd through oracle/torch_stock.py + SiSNRWavLossTorch, in
    fp64) reproduces what the REFERENCE's own training step produced in fp64: loss (ss_losses.py:21-26,96-130) and
    d loss / d every parameter after loss.backward() (trainer.py:40-47), captured by tools/gen_golden.py from the imported
    reference.  In fp32 the port's loss equals the reference's fp32 loss."""
    import torch
    from oracle.torch_stock import SiSNRWavLossTorch, StockDPTN
    cfg, z = golden(name)
    B, T, Tv = (int(v) for v in z["shape"])
    wseed, iseed = (int(v) for v in z["seeds"])
    sd = synthetic_state_dict(cfg, seed=wseed)
    assert weights_digest(sd) == str(z["digest"])
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=iseed)
    out32 = StockDPTN(cfg, sd)(**{k: torch.from_numpy(v) for k, v in inp.items()})
    loss32 = SiSNRWavLossTorch()(**{**{k: torch.from_numpy(v) for k, v in inp.items()}, **out32})["loss"]
    assert abs(float(loss32) - float(z["val.loss"])) < 1e-5 * abs(float(z["val.loss"]))
    for k in ("s1_pred", "s2_pred"):        # train mode with dropout 0 == eval mode
        assert O.agreement_db(out32[k].numpy(), z["tap." + k]) > 100
    ref = StockDPTN(cfg, sd)
    ref.sd = {k: v.double().requires_grad_(True) for k, v in ref.sd.items()}
    ref.paths = [(pre, m.double() if m is not None else None, r.double()) for pre, m, r in ref.paths]
    for _, m, r in ref.paths:
        for p in (list(m.parameters()) if m is not None else []) + list(r.parameters()):
            p.requires_grad_(True)
    batch = {k: torch.from_numpy(v).double() for k, v in inp.items()}
    with torch.enable_grad():
        batch.update(StockDPTN.__call__.__wrapped__(ref, **batch))
        loss = SiSNRWavLossTorch()(**batch)["loss"]
        loss.backward()
    assert abs(float(loss.detach()) - float(z["val.loss64"])) < 1e-9 * abs(float(z["val.loss64"]))
    grads = {}
    for pre, m, r in ref.paths:
        if m is not None:
            grads[pre + "mha.in_proj_weight"], grads[pre + "mha.in_proj_bias"] = m.in_proj_weight.grad, m.in_proj_bias.grad
            grads[pre + "mha.out_proj.weight"], grads[pre + "mha.out_proj.bias"] = m.out_proj.weight.grad, m.out_proj.bias.grad
        for k, v in r.named_parameters():
            grads[pre + "rnn." + k] = v.grad
    for k, v in ref.sd.items():
        if k not in grads:
            grads[k] = v.grad
    fails, worst_db, worst_norm = reference_gradient_report(z, {k: v.numpy() for k, v in grads.items()}, floor_db=120.0,
                                                            margin_db=-1e9)
    assert not fails, fails[:5]
    assert worst_norm[0] < 1e-7, worst_norm
