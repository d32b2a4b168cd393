"""Seeded random configurations of the whole forward against the fp64 oracle (GPU box).

The hand-picked cases of test_gpu_parity.py cover the shapes the BASELINE configurations use; the two latent bugs found in
round 3 (a fragment batch that did not tile K = 192, a weight pack that read a 128 x 128 matrix as 128 x 256) sat in shapes
nobody had picked.  This file draws architecture, feature count, chunking, encoder kernel, directions, batch, clip length
and video length at random (fixed seeds: a regular, reproducible test) and compares the outputs of dptnav_forward with the
numpy restatement of the reference in float64 (oracle/dptn_oracle.py, pinned to the reference by test_oracle_golden.py),
once per kernel selection the sizes allow.  Sizes are kept where the oracle takes about a second.
"""
import numpy as np
import pytest
import torch

from oracle import dptn_oracle as O
from speech_separation_amd.spec import DPTNConfig, synthetic_inputs, synthetic_state_dict

pytestmark = pytest.mark.gpu
MIN_DB = 80.0


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    arch = "dptn" if rng.random() < 0.55 else "dprnn"
    N = int(rng.choice([64, 128]))
    K = int(rng.choice([20, 50, 64, 100, 150, 200, 250]))
    kenc = int(rng.choice([2, 4, 5, 7, 8]))
    audio_only = bool(rng.random() < 0.4)
    cfg = DPTNConfig(num_features=N, hidden_video=N, kernel_size_enc=kenc, hidden_dim=128, num_blocks=int(rng.choice([1, 2])),
                     chunk_size=K, step_size=K // 2, num_heads=4, bidir=bool(rng.random() < 0.75), audio_only=audio_only,
                     arch=arch)
    B = int(rng.integers(1, 6))
    stride = kenc // 2
    S = int(rng.integers(1, 9))                                   # chunks
    L = (S - 1) * (K // 2) + K + int(rng.integers(0, K // 2))     # frames: S chunks + a trailing remainder the fold drops
    T = (L - 1) * stride + kenc + int(rng.integers(0, stride))    # samples: L frames + a remainder the encoder drops
    Tv = 1 if audio_only else int(rng.choice([1, 3, 7, 25, 50, 61]))
    return cfg, B, T, Tv


@pytest.mark.parametrize("seed", range(32))
def test_random_configuration_matches_the_oracle(seed):
    from speech_separation_amd.engine import DptnEngine, params_to_device
    dev = torch.device("cuda:0")
    cfg, B, T, Tv = draw(seed)
    sd = synthetic_state_dict(cfg, seed=seed)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=seed)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    ref = O.forward(cfg, sd, dtype=np.float64, **inp)
    args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    what = f"seed {seed}: {cfg.arch} N={cfg.num_features} K={cfg.chunk_size} kenc={cfg.kernel_size_enc} bidir={cfg.bidir} " \
           f"audio_only={cfg.audio_only} blocks={cfg.num_blocks} B={B} T={T} Tv={Tv}"
    worst = 1e9
    # kernel selections: default (low-latency recurrence + fused blocks where they apply), 16-sequence tiles with the input
    # projection inside the recurrence (lstm16x.hip; 2 = also for 128 features below 12 mixtures), the same tiles fed by the
    # K4 GEMM, 32-sequence tiles with separate attention launches, one stream
    for opts in ({}, {"lstm4": 0, "fuse_pre128": 2}, {"lstm4": 0, "fuse_pre": 0, "fuse_pre128": 0}, {"lstm16": 0, "fuse_attn": 0},
                 {"overlap": 0, "fuse_ffn": 0}):
        for k, v in {"lstm4": 1, "lstm16": 1, "fuse_attn": 1, "fuse_ffn": 1, "overlap": 1, "fuse_pre": 1, "fuse_pre128": 1, **opts}.items():
            eng.set_option(k, v)
        s1, s2 = eng.forward(*args)
        torch.cuda.synchronize()
        assert s1.shape == (B, T) and s2.shape == (B, T), what
        for got, key in ((s1, "s1_pred"), (s2, "s2_pred")):
            g = got.cpu().numpy()
            assert np.isfinite(g).all(), (what, opts)
            db = O.agreement_db(g, ref[key])
            worst = min(worst, db)
            assert db > MIN_DB, (what, opts, key, db)
    print(f"{what}: worst {worst:.1f} dB over 5 kernel selections")


def stock_gradients(cfg, sd, inp, d1, d2):
    """fp64 autograd through the stock-PyTorch composition (oracle/torch_stock.py, pinned to the reference's own fp64
    gradients by tests/test_oracle_golden.py) for loss = <s1_pred, d1> + <s2_pred, d2>."""
    from oracle.torch_stock import StockDPTN
    ref = StockDPTN(cfg, sd)
    ref.sd = {k: v.double().requires_grad_(True) for k, v in ref.sd.items()}
    ref.paths = [(pre, m.double() if m is not None else None, r.double()) for pre, m, r in ref.paths]
    for _, m, r in ref.paths:
        for p in (list(m.parameters()) if m is not None else []) + list(r.parameters()):
            p.requires_grad_(True)
    # smallest |h| that meets the FFN's ReLU (dptn.py:31): below ~1e-7 fp32 and fp64 arithmetic disagree on its SIGN, the
    # ReLU's derivative flips for that element and one row of the LSTM gradients comes out at ~48 dB with every kernel right
    smallest = [np.inf]
    rnn_forwards = []
    for _, m, r in ref.paths:
        if m is not None:
            rnn_forwards.append(r.register_forward_hook(lambda mod, a, out: smallest.__setitem__(0, min(smallest[0], float(out[0].detach().abs().min())))))
    with torch.enable_grad():
        out = StockDPTN.__call__.__wrapped__(ref, **{k: torch.from_numpy(v).double() for k, v in inp.items()})
        (out["s1_pred"] * torch.from_numpy(d1).double() + out["s2_pred"] * torch.from_numpy(d2).double()).sum().backward()
    for hnd in rnn_forwards:
        hnd.remove()
    want = {"__smallest_relu_input__": smallest[0]}
    for pre, m, r in ref.paths:
        if m is not None:
            want[pre + "mha.in_proj_weight"], want[pre + "mha.in_proj_bias"] = m.in_proj_weight.grad, m.in_proj_bias.grad
            want[pre + "mha.out_proj.weight"], want[pre + "mha.out_proj.bias"] = m.out_proj.weight.grad, m.out_proj.bias.grad
        for k, v in r.named_parameters():
            want[pre + "rnn." + k] = v.grad
    for k, v in ref.sd.items():
        if k not in want and v.grad is not None:
            want[k] = v.grad
    return want


@pytest.mark.parametrize("seed", range(100, 112))
def test_random_configuration_gradients_match_autograd(seed):
    """The training step's backward (dptnav_train_forward / dptnav_train_backward) for random configurations: every
    parameter gradient against fp64 autograd, both recurrence / BPTT tile heights."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    dev = torch.device("cuda:0")
    cfg, B, T, Tv = draw(seed)
    cfg = DPTNConfig(**{**cfg.to_dict(), "dropout": 0.0})
    B = min(B, 3)
    sd = synthetic_state_dict(cfg, seed=seed)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=seed)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    rng = np.random.default_rng(seed)
    d1 = rng.standard_normal((B, T)).astype(np.float32)
    d2 = rng.standard_normal((B, T)).astype(np.float32)
    want = stock_gradients(cfg, sd, inp, d1, d2)
    kink = want.pop("__smallest_relu_input__")
    what = f"seed {seed}: {cfg.arch} N={cfg.num_features} K={cfg.chunk_size} kenc={cfg.kernel_size_enc} bidir={cfg.bidir} " \
           f"audio_only={cfg.audio_only} blocks={cfg.num_blocks} B={B} T={T} Tv={Tv}"
    args = (t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    for tile in (16, 32):
        eng.set_option("lstm16", 1 if tile == 16 else 0)
        s1, s2, tape = eng.train_forward(*args)
        eng.train_backward(*args, torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev), tape)
        torch.cuda.synchronize()
        missing = [k for k in grads if k not in want]
        assert not missing, (what, missing)
        worst = min((O.agreement_db(grads[k].cpu().numpy(), want[k].numpy().reshape(grads[k].shape)), k) for k in grads)
        print(f"{what} tile {tile}: worst parameter {worst[1]} {worst[0]:.1f} dB (smallest ReLU input {kink:.1e})")
        if worst[0] <= 60 and kink < 1e-7 and ".rnn." in worst[1]:
            # (seed 102: h = +1.8e-8 in fp64, -5.3e-8 in torch's own fp32 forward; 32-sequence tiles land on the other side)
            pytest.skip(f"{what}: a ReLU input of {kink:.1e} (fp64) -- its sign is not defined in fp32 arithmetic and one kernel "
                        f"selection landed on the other side ({worst[1]}: {worst[0]:.1f} dB); nothing to compare")
        assert worst[0] > 60, (what, tile, worst)
