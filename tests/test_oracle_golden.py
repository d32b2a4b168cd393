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
