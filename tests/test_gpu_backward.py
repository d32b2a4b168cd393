"""Training-step parity (GPU): the path-level backward of libdptnav against torch.autograd on the stock-PyTorch CPU
composition (oracle/torch_stock.py) -- the same graph the reference's loss.backward() differentiates (dptn.py:36-52)."""
import numpy as np
import pytest
import torch

from oracle import dptn_oracle as O
from oracle.torch_stock import StockDPTN
from speech_separation_amd.spec import DPTN_AV, DPTNConfig, synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("features", [128, 64])
@pytest.mark.parametrize("path", [0, 1])
def test_path_backward_matches_autograd(dev, path, features):
    """One TransformerDPRNN, forward with tape + backward, against fp64 autograd: num_features = 128 (DPTNAVWavEncDec, the
    tuned kernels) and 64 (DPTNWavEncDec: generic weight-gradient kernel, 64 x 64 data-gradient tiles)."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 1, "dropout": 0.0, "num_features": features,
                        "audio_only": features == 64})
    sd = synthetic_state_dict(cfg, seed=4)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    B, S, K, N = 2, 3, cfg.chunk_size, cfg.num_features
    # (input seed: the FFN's ReLU has a kink at h = 0, and with 10^5 hidden values per direction the smallest |h| of a
    #  draw is 1e-8..1e-6 -- below ~3e-8 fp32 and fp64 disagree on its sign and the gradient of that one unit differs by a
    #  finite amount: seen with seed 1 at 64 features (|h| = 2.5e-8, one row of the LSTM gradients at 48 dB).  Seed 25
    #  keeps every |h| above 4e-7 there; tools/ has no part in this, it is a property of the function)
    rng = np.random.default_rng(1 if features == 128 else 25)
    x = rng.standard_normal((B, S, K, N)).astype(np.float32)
    dy = rng.standard_normal((B, S, K, N)).astype(np.float32)

    # ---- ours ----
    xt = torch.from_numpy(x).to(dev)
    y, tape = eng.train_path_forward(0, path, xt)
    y_inf = eng.stage_path(0, path, xt)
    assert O.agreement_db(y.cpu().numpy(), y_inf.cpu().numpy()) > 120          # training forward == inference forward
    dx = eng.train_path_backward(0, path, xt, torch.from_numpy(dy).to(dev), tape)
    torch.cuda.synchronize()

    # ---- torch autograd on CPU (float64 for a tight reference) ----
    ref = StockDPTN(cfg, sd)
    name = "intra_chunk_block" if path == 0 else "inter_chunk_block"
    pre = f"dprnn.model.0.{name}."
    _, mha, rnn = ref.paths[path]
    mha, rnn = mha.double().train(False), rnn.double()
    params = {k: v.double().requires_grad_(True) for k, v in ref.sd.items() if k.startswith(pre)}
    ref.sd.update(params)
    for p in list(mha.parameters()) + list(rnn.parameters()):
        p.requires_grad_(True)
    xs = torch.from_numpy(x).double()
    seqs = (xs.reshape(B * S, K, N) if path == 0 else xs.transpose(1, 2).reshape(B * K, S, N)).requires_grad_(True)
    with torch.enable_grad():
        out = ref._path.__wrapped__(ref, seqs, pre, mha, rnn) if hasattr(ref._path, "__wrapped__") else ref._path(seqs, pre, mha, rnn)
        dys = torch.from_numpy(dy).double()
        dseq = dys.reshape(B * S, K, N) if path == 0 else dys.transpose(1, 2).reshape(B * K, S, N)
        out.backward(dseq)
    want_dx = seqs.grad.reshape(B, S, K, N) if path == 0 else seqs.grad.reshape(B, K, S, N).transpose(1, 2)
    assert O.agreement_db(dx.cpu().numpy(), want_dx.numpy()) > 80, "d x"

    want = {"mha.in_proj_weight": mha.in_proj_weight.grad, "mha.in_proj_bias": mha.in_proj_bias.grad,
            "mha.out_proj.weight": mha.out_proj.weight.grad, "mha.out_proj.bias": mha.out_proj.bias.grad}
    for k, v in rnn.named_parameters():
        want["rnn." + k] = v.grad
    for leaf in ("ln1.weight", "ln1.bias", "ffn.1.weight", "ffn.1.bias", "ln2.weight", "ln2.bias"):
        want[leaf] = params[pre + leaf].grad
    for leaf, gref in want.items():
        got = grads[pre + leaf].cpu().numpy()
        assert got.shape == tuple(gref.shape), leaf
        assert O.agreement_db(got, gref.numpy()) > 70, (leaf, O.agreement_db(got, gref.numpy()))


@pytest.mark.parametrize("lstm_tile", [16, 32])     # both recurrence / BPTT kernel pairs
@pytest.mark.parametrize("audio_only,features,bidir", [(False, 128, True), (True, 128, True), (True, 64, True), (False, 128, False)],
                         ids=["av128", "audio128", "audio64", "av128_unidir"])
def test_whole_model_backward_matches_autograd(dev, audio_only, features, bidir, lstm_tile):
    """d loss / d every parameter for a 2-block model: libdptnav train_forward/backward vs torch.autograd (fp64, CPU).
    audio64 = the reference's DPTNWavEncDec configuration (model/dptn_wav.yaml: 64 features); av128_unidir = bidir False
    (dptn.py:60): its inter-chunk paths keep their LSTM weight gradients in the half's stream while the intra-chunk ones
    go to the side stream (the dP buffer hand-over between the two was a race until round 4, tests/test_gpu_memsafety.py)."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2, "dropout": 0.0, "audio_only": audio_only, "num_features": features,
                        "bidir": bidir})
    sd = synthetic_state_dict(cfg, seed=2)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    eng.set_option("lstm16", 1 if lstm_tile == 16 else 0)
    B, T, Tv = 2, 2000, 9
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=6)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    rng = np.random.default_rng(3)
    d1 = rng.standard_normal((B, T)).astype(np.float32)
    d2 = rng.standard_normal((B, T)).astype(np.float32)
    s1, s2, tape = eng.train_forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    f1, f2 = eng.forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
    assert O.agreement_db(s1.cpu().numpy(), f1.cpu().numpy()) > 110 and O.agreement_db(s2.cpu().numpy(), f2.cpu().numpy()) > 110
    eng.train_backward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"), torch.from_numpy(d1).to(dev),
                       torch.from_numpy(d2).to(dev), tape)
    torch.cuda.synchronize()

    ref = StockDPTN(cfg, sd)
    ref.sd = {k: v.double().requires_grad_(True) for k, v in ref.sd.items()}
    ref.paths = [(pre, m.double() if m is not None else None, r.double()) for pre, m, r in ref.paths]
    for _, m, r in ref.paths:
        for p in list(m.parameters()) + list(r.parameters()):
            p.requires_grad_(True)
    with torch.enable_grad():
        out = StockDPTN.__call__.__wrapped__(ref, **{k: torch.from_numpy(v).double() for k, v in inp.items()})
        (out["s1_pred"] * torch.from_numpy(d1).double() + out["s2_pred"] * torch.from_numpy(d2).double()).sum().backward()
    want = {}
    for pre, m, r in ref.paths:
        want[pre + "mha.in_proj_weight"], want[pre + "mha.in_proj_bias"] = m.in_proj_weight.grad, m.in_proj_bias.grad
        want[pre + "mha.out_proj.weight"], want[pre + "mha.out_proj.bias"] = m.out_proj.weight.grad, m.out_proj.bias.grad
        for k, v in r.named_parameters():
            want[pre + "rnn." + k] = v.grad
    for k, v in ref.sd.items():
        if k not in want and v.grad is not None:
            want[k] = v.grad
    missing = [k for k in grads if k not in want]
    assert not missing, missing
    worst = min((O.agreement_db(grads[k].cpu().numpy(), want[k].numpy().reshape(grads[k].shape)), k) for k in grads)
    assert worst[0] > 60, worst


def test_training_steps_match_stock_pytorch(dev):
    """Three optimizer steps (PIT SI-SNR loss, clip 10, AdamW 1e-3 = src/configs/dptn_wav_av.yaml:9-11,25) through the
    nn.Module drop-in with the device loss / clip / FusedAdamW vs the same steps on the stock-PyTorch CPU composition with
    torch.autograd, the torch-operator loss (oracle, pinned to the reference's values), torch's clip and torch.optim.AdamW."""
    from oracle.torch_stock import SiSNRWavLossTorch
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.spec import synthetic_inputs
    from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step
    kw = dict(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128, num_blocks=1,
              chunk_size=150, step_size=75, dropout=0.0, num_heads=4, bidir=True)
    model = DPTNAVWavEncDec(**kw)
    cfg = model.cfg
    sd = synthetic_state_dict(cfg, seed=5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    inp = synthetic_inputs(cfg, B=2, T=2000, Tv=50, seed=8)
    ours = []
    for _ in range(3):
        batch = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
        ours.append({k: float(v) for k, v in train_step(model, batch, SiSNRWavLoss(), opt, max_grad_norm=10.0).items()})
    # autograd adopted the per-parameter views of the step's ONE flat gradient copy (no 228 copies, one collective)
    flat = model._flat_grad
    assert all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in model.parameters())

    # the same three steps with stock PyTorch on the CPU
    class Stock(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ref = StockDPTN(cfg, sd)
            self.p = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone()) for k, v in self.ref.sd.items()
                                             if ".mha." not in k and ".rnn." not in k})
            self.mods = torch.nn.ModuleList([m for _, m, _ in self.ref.paths] + [r for _, _, r in self.ref.paths])
            for k in self.p:
                self.ref.sd[k.replace("/", ".")] = self.p[k]

        def forward(self, **b):
            return StockDPTN.__call__.__wrapped__(self.ref, **b)
    stock = Stock()
    sopt = torch.optim.AdamW(stock.parameters(), lr=1e-3)
    theirs = []
    for _ in range(3):      # trainer.py:38-51 with stock operators
        batch = {k: torch.from_numpy(v) for k, v in inp.items()}
        sopt.zero_grad()
        batch.update(stock(**batch))
        loss = SiSNRWavLossTorch()(**batch)["loss"]
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(stock.parameters(), 10.0)
        sopt.step()
        theirs.append({"loss": float(loss), "grad_norm": float(norm)})
    for a, b in zip(ours, theirs):
        assert abs(a["loss"] - b["loss"]) < 2e-3 * max(1.0, abs(b["loss"])), (ours, theirs)
        assert abs(a["grad_norm"] - b["grad_norm"]) < 1e-3 * b["grad_norm"], (ours, theirs)
    assert ours[2]["loss"] < ours[0]["loss"]          # and the loss goes down


@pytest.mark.parametrize("path", [0, 1])
def test_dropout_forward_and_backward_match_autograd_with_the_same_mask(dev, path):
    """Train-mode attention dropout: the kernels' counter-based keep-mask is materialised (dptnav_dropout_mask) and fed to
    an explicit-formula torch implementation of the path; outputs, dx and all parameter gradients must agree."""
    import torch.nn.functional as F
    from speech_separation_amd.engine import DptnEngine, params_to_device
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 1, "dropout": 0.1})
    sd = synthetic_state_dict(cfg, seed=9)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    eng.set_option("dropout_ppm", 100000)
    eng.set_option("dropout_seed", 12345)
    B, S, K, N, H, heads = 2, 3, cfg.chunk_size, cfg.num_features, cfg.hidden_dim, cfg.num_heads
    rng = np.random.default_rng(2)
    x = rng.standard_normal((B, S, K, N)).astype(np.float32)
    dy = rng.standard_normal((B, S, K, N)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    y, tape = eng.train_path_forward(0, path, xt)
    dx = eng.train_path_backward(0, path, xt, torch.from_numpy(dy).to(dev), tape)
    mask = eng.dropout_mask(0, path, B, S).cpu().double()
    torch.cuda.synchronize()
    keep = float(mask.mean())
    assert 0.88 < keep < 0.92, keep                       # p = 0.1
    # integer work: the device mask equals the numpy restatement of the counter-based generator bit for bit
    from dropout_ref import keep_mask
    assert np.array_equal(mask.numpy().astype(np.float32), keep_mask(0, path, B, S, K, heads, 100000, 12345))
    eng.set_option("dropout_seed", 54321)
    assert not torch.equal(eng.dropout_mask(0, path, B, S).cpu().double(), mask)   # the seed matters

    name = "intra_chunk_block" if path == 0 else "inter_chunk_block"
    pre = f"dprnn.model.0.{name}."
    P = {k[len(pre):]: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
    rnn = torch.nn.LSTM(N, H, bidirectional=True, batch_first=True).double()
    rnn.load_state_dict({k[4:]: v.detach() for k, v in P.items() if k.startswith("rnn.")})
    xs = torch.from_numpy(x).double()
    seqs = (xs.reshape(B * S, K, N) if path == 0 else xs.transpose(1, 2).reshape(B * K, S, N)).requires_grad_(True)
    R, Ls = seqs.shape[0], seqs.shape[1]
    dh = N // heads
    with torch.enable_grad():
        qkv = F.linear(seqs, P["mha.in_proj_weight"], P["mha.in_proj_bias"])
        q, k, v = (t.reshape(R, Ls, heads, dh).transpose(1, 2) for t in qkv.split(N, -1))
        prob = torch.softmax(q @ k.transpose(-1, -2) / dh ** 0.5, -1) * mask / 0.9
        att = (prob @ v).transpose(1, 2).reshape(R, Ls, N)
        y1 = F.layer_norm(F.linear(att, P["mha.out_proj.weight"], P["mha.out_proj.bias"]) + seqs, (N,), P["ln1.weight"], P["ln1.bias"])
        z = F.linear(F.relu(rnn(y1)[0]), P["ffn.1.weight"], P["ffn.1.bias"]) + y1
        out = F.layer_norm(z, (N,), P["ln2.weight"], P["ln2.bias"])
        dys = torch.from_numpy(dy).double()
        out.backward(dys.reshape(B * S, K, N) if path == 0 else dys.transpose(1, 2).reshape(B * K, S, N))

    def back(a):
        return a.reshape(B, S, K, N) if path == 0 else a.reshape(B, K, S, N).transpose(1, 2)
    assert O.agreement_db(y.cpu().numpy(), back(out.detach()).numpy()) > 80
    assert O.agreement_db(dx.cpu().numpy(), back(seqs.grad).numpy()) > 70
    want = {k: v.grad for k, v in P.items() if not k.startswith("rnn.")}
    want.update({"rnn." + k: v.grad for k, v in rnn.named_parameters()})
    for leaf, gref in want.items():
        assert O.agreement_db(grads[pre + leaf].cpu().numpy(), gref.numpy()) > 65, leaf


def test_split_training_step_equals_unsplit(dev):
    """The training step runs a batch as two halves on two streams (option train_overlap).  With an odd batch (halves of
    2 and 1 mixtures; dropout off, the halves draw different masks by design) outputs and every parameter gradient
    must equal the unsplit step's up to the fp32 summation order of the two halves' contributions."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2, "dropout": 0.0})
    sd = synthetic_state_dict(cfg, seed=4)
    B, T, Tv = 3, 2300, 11
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=10)
    rng = np.random.default_rng(11)
    d1 = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32)).to(dev)
    d2 = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32)).to(dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    results = []
    for split in (1, 0):
        eng = DptnEngine(cfg, dev)
        eng.bind(params_to_device(sd, dev))
        grads = eng.bind_grads()
        eng.set_option("train_overlap", split)
        s1, s2, tape = eng.train_forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
        eng.train_backward(t["mix"], t["s1_embedding"], t["s2_embedding"], d1, d2, tape)
        torch.cuda.synchronize()
        results.append((s1.cpu().numpy(), s2.cpu().numpy(), {k: g.cpu().numpy().copy() for k, g in grads.items()}))
    (a1, a2, ga), (b1, b2, gb) = results
    assert O.agreement_db(a1, b1) > 110 and O.agreement_db(a2, b2) > 110
    worst = min((O.agreement_db(ga[k], gb[k]), k) for k in ga)
    assert worst[0] > 90, worst


def test_weight_gradient_schedules_agree(dev):
    """Where the weight gradients are formed is a schedule, not arithmetic: riding on the data-gradient GEMMs (option
    wgrad_ride), on a side stream per half with rotating dP buffers (wgrad_side), both in one pass over dP (wgrad2) -- or
    as stand-alone launches in the half's own stream.  Four paths deep so that every dP buffer is reused; every
    gradient must agree with the plain schedule's to fp32 summation order, and with dropout on (same seed) too."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 2})
    sd = synthetic_state_dict(cfg, seed=6)
    B, T, Tv = 4, 2600, 13
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=12)
    rng = np.random.default_rng(13)
    d1 = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32)).to(dev)
    d2 = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32)).to(dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    results = {}
    for name, opts in {"default": {}, "plain": {"wgrad_ride": 0, "wgrad_side": 0, "wgrad2": 0, "ln_tape": 0},
                       "no side stream": {"wgrad_side": 0}, "no riders": {"wgrad_ride": 0},
                       "ln recompute": {"ln_tape": 0}, "chained recurrences": {"lstm_chain": 1}}.items():
        eng = DptnEngine(cfg, dev)
        eng.bind(params_to_device(sd, dev))
        grads = eng.bind_grads()
        eng.set_option("dropout_ppm", 100000)
        eng.set_option("dropout_seed", 99)
        for k, v in opts.items():
            eng.set_option(k, v)
        for _ in range(2):      # twice: the second backward reuses events, buffers and ticket counters of the first
            s1, s2, tape = eng.train_forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
            eng.train_backward(t["mix"], t["s1_embedding"], t["s2_embedding"], d1, d2, tape)
        torch.cuda.synchronize()
        results[name] = {k: g.cpu().numpy().copy() for k, g in grads.items()}
        del eng
    for name in ("default", "no side stream", "no riders", "ln recompute", "chained recurrences"):
        worst = min((O.agreement_db(results[name][k], results["plain"][k]), k) for k in results["plain"])
        assert worst[0] > 90, (name, worst)


@pytest.mark.parametrize("deterministic", [0, 1])
def test_full_size_training_step_is_deterministic(dev, deterministic):
    """B=16, T=32000 (BASELINE config 4, dropout 0.1): two forward/backward passes with the same dropout seed give
    bit-identical outputs; the parameter gradients agree to fp32 summation order (the token reductions hand their tiles
    out by dynamic tickets, so WHICH workgroup sums which tiles -- not the values summed -- varies from run to run; no
    float atomics anywhere), and everything is finite.  With option deterministic = 1 (static tile order) the gradients
    are bit-identical too, and equal to the default's up to that summation order."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTN_AV
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, seed=0), dev))
    grads = eng.bind_grads()
    eng.set_option("dropout_ppm", 100000)
    eng.set_option("dropout_seed", 2024)
    eng.set_option("deterministic", deterministic)
    B, T = 16, 32000
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=3)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    g = torch.Generator(device="cpu").manual_seed(0)
    d1 = torch.randn(B, T, generator=g).to(dev)
    d2 = torch.randn(B, T, generator=g).to(dev)
    runs = []
    for _ in range(2):
        s1, s2, tape = eng.train_forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
        eng.train_backward(t["mix"], t["s1_embedding"], t["s2_embedding"], d1, d2, tape)
        torch.cuda.synchronize()
        runs.append((s1.clone(), s2.clone(), eng._grads_flat.clone()))
        del tape
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    if deterministic:
        assert torch.equal(runs[0][2], runs[1][2]), "static tile order: the gradients must be bit-identical run to run"
        eng.set_option("deterministic", 0)          # ... and the ticket-driven default computes the same sums in another order
        s1, s2, tape = eng.train_forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
        eng.train_backward(t["mix"], t["s1_embedding"], t["s2_embedding"], d1, d2, tape)
        torch.cuda.synchronize()
        assert torch.equal(s1, runs[0][0])
        runs[1] = (s1.clone(), s2.clone(), eng._grads_flat.clone())
        del tape
    ga, gb = runs[0][2].double(), runs[1][2].double()
    db = 10 * torch.log10(ga.pow(2).sum() / (ga - gb).pow(2).sum().clamp_min(1e-300))
    assert float(db) > 110, float(db)
    # per parameter as well (a race would show up as a large error in one tensor)
    for k, g in eng._grads.items():
        o = eng._grad_offsets[k]
        a, b = ga[o:o + g.numel()], gb[o:o + g.numel()]
        if float(a.pow(2).sum()) > 0:
            assert float(10 * torch.log10(a.pow(2).sum() / (a - b).pow(2).sum().clamp_min(1e-300))) > 90, k
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][2]).all()
    assert float(runs[0][2].abs().max()) > 0.0


@pytest.mark.parametrize("lstm_tile", [16, 32])     # both recurrence / BPTT kernel pairs
@pytest.mark.parametrize("name,copies", [("grad_mid_av", 1), ("grad_full_av", 1), ("grad_full_av", 16), ("grad_mid_audio", 1),
                                         ("grad_mid_dprnn", 1)])
def test_training_step_matches_reference_gradients(dev, golden, name, copies, lstm_tile):
    """BASELINE config 4 against the REFERENCE's own numbers: tests/golden/grad_*.npz hold the loss and d loss / d every
    parameter that the imported reference produced with `model.train(); outputs = model(**batch); SiSNRWavLoss;
    loss.backward()` (trainer.py:40-47, ss_losses.py:21-26,96-130, dptn_wav.py:171-194; dropout 0.0 -- tools/gen_golden.py).
    Here the same step runs through the drop-in nn.Module (dptnav_train_forward / dptnav_train_backward) and the device
    loss (dptnav_pit_sisnr_loss); the truth is the same reference step run in fp64.  `copies` = 16 repeats the fixture's one mixture over the batch of config 4
    (B=16 x T=32000, 6 blocks): the batch mean of 16 identical terms is the single term, so loss and gradients must
    reproduce the B=1 reference numbers while every kernel runs at its full BASELINE size (two halves, two streams)."""
    from speech_separation_amd import DPRNNEncDec, DPTNAVWavEncDec, DPTNWavEncDec
    from speech_separation_amd.spec import synthetic_inputs
    from speech_separation_amd.train import SiSNRWavLoss
    from tests.test_oracle_golden import reference_gradient_report
    from tools.gen_golden import weights_digest
    cfg, z = golden(name)
    B, T, Tv = (int(v) for v in z["shape"])
    wseed, iseed = (int(v) for v in z["seeds"])
    sd = synthetic_state_dict(cfg, seed=wseed)
    assert weights_digest(sd) == str(z["digest"])
    kw = {k: v for k, v in cfg.to_dict().items() if k not in ("audio_only", "arch")}
    if cfg.arch == "dprnn":      # grad_mid_dprnn: the reference's DPRNNEncDec (dprnn.py:230-289)
        for k in ("video_emb_size", "hidden_video", "num_heads", "dropout"):
            kw.pop(k)
        model = DPRNNEncDec(**kw)
    elif cfg.audio_only:         # grad_mid_audio: the reference's DPTNWavEncDec (dptn_wav.py:64-126), 64 features
        for k in ("video_emb_size", "hidden_video"):
            kw.pop(k)
        model = DPTNWavEncDec(**kw)
    else:
        model = DPTNAVWavEncDec(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).train()
    model._get_engine(dev).set_option("lstm16", 1 if lstm_tile == 16 else 0)
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=iseed)
    batch = {k: torch.from_numpy(np.repeat(v, copies, axis=0)).to(dev) for k, v in inp.items()}
    batch.update(model(mix_spectrogram=torch.zeros(1, device=dev), **batch))
    loss = SiSNRWavLoss()(**batch)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(z["val.loss64"])) < 1e-5 * abs(float(z["val.loss64"])), (float(loss), float(z["val.loss64"]))
    for k in ("s1_pred", "s2_pred"):        # the training forward reproduces the reference's train-mode outputs, every row
        for r in range(0, B * copies, B):
            assert O.agreement_db(batch[k][r:r + B].detach().cpu().numpy(), z["tap." + k]) > 80, (k, r)
    grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
    # truth = the reference's step run in fp64; a parameter must agree to 60 dB, or to within 10 dB (a factor 3 in error
    # amplitude between two fp32 evaluation orders) of what the reference's OWN fp32 step achieves for it: with random
    # weights the SI-SNR projection <p, g> / |g|^2 is a small difference of large sums, so every gradient of the fp32 step
    # carries a common relative error of ~3e-3 (50-56 dB) -- in the reference's arithmetic as much as in ours
    fails, worst_db, worst_norm = reference_gradient_report(z, grads, floor_db=60.0, margin_db=10.0)
    print(f"{name} x{copies} tile {lstm_tile}: worst parameter {worst_db[1]} {worst_db[0]:.1f} dB (the reference's fp32 step: "
          f"{float(z['ref32db.' + worst_db[1]]):.1f} dB), worst norm error {worst_norm[0]:.2e} ({worst_norm[1]})")
    assert not fails, fails[:8]
    assert worst_norm[0] < 3e-3, worst_norm
    total = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    assert abs(total - float(z["val.grad_norm"])) < 1e-4 * float(z["val.grad_norm"])
    # the fused clip sees the same global norm the reference's clip_grad_norm_ would (base_trainer.py:383-391)
    from speech_separation_amd.train import clip_grad_norm_
    norm = clip_grad_norm_(model, 10.0)
    assert abs(float(norm) - float(z["val.grad_norm"])) < 1e-4 * float(z["val.grad_norm"])


def test_backward_refuses_a_tape_written_under_the_measurement_probe(dev):
    """Option train_fuse_probe (tools/train_fuse_probe.py) makes the training forward run the inference attention block: no
    qkv / attention / LayerNorm tape, no dropout.  A backward behind it used to return OK with garbage gradients; both
    backward entry points now refuse while the option is set, and work again once it is cleared."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 1})
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, seed=0), dev))
    eng.bind_grads()
    t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=2, T=4000, Tv=50, seed=3).items()}
    args = (t["mix"], t["s1_embedding"], t["s2_embedding"])
    d = torch.randn(2, 4000, device=dev)
    eng.set_option("train_fuse_probe", 1)
    s1, s2, tape = eng.train_forward(*args)
    with pytest.raises(RuntimeError, match="train_fuse_probe"):
        eng.train_backward(*args, d, d, tape)
    S = eng.chunks(4000)
    x = torch.randn(2, S, cfg.chunk_size, cfg.num_features, device=dev)
    y, ptape = eng.train_path_forward(0, 0, x)
    with pytest.raises(RuntimeError, match="train_fuse_probe"):
        eng.train_path_backward(0, 0, x, torch.randn_like(x), ptape)
    eng.set_option("train_fuse_probe", 0)
    s1, s2, tape = eng.train_forward(*args)
    eng.train_backward(*args, d, d, tape)
    torch.cuda.synchronize()
    assert all(torch.isfinite(g).all() for g in eng._grads.values())


def test_training_rejects_long_video_before_launching_anything(dev):
    """the training step's limit of 256 video frames is reported by the size queries / dptnav_train_forward,
    not by the last stage of the backward after the whole step has run; inference takes the same clip."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import synthetic_inputs
    cfg = DPTNConfig(**{**DPTN_AV.to_dict(), "num_blocks": 1, "dropout": 0.0})
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, seed=1), dev))
    eng.bind_grads()
    inp = synthetic_inputs(cfg, B=2, T=2000, Tv=300, seed=2)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    with pytest.raises(RuntimeError, match="at most 256 video frames"):
        eng.train_forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
    s1, s2 = eng.forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
    torch.cuda.synchronize()
    assert torch.isfinite(s1).all() and torch.isfinite(s2).all()


@pytest.mark.parametrize("features", [64, 128])
@pytest.mark.parametrize("path", [0, 1])
def test_dprnn_path_backward_matches_autograd(dev, path, features):
    """One DPRNN block half (IntraChunkRNN / InterChunkRNN, dprnn.py:24-47,65-89: bi-LSTM -> fc -> LayerNorm -> + x) with
    tape + backward against fp64 autograd on the stock composition: d x and the 12 parameter gradients."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import DPRNN_AUDIO
    cfg = DPTNConfig(**{**DPRNN_AUDIO.to_dict(), "num_blocks": 1, "num_features": features, "hidden_video": features,
                        "chunk_size": 50, "step_size": 25})
    sd = synthetic_state_dict(cfg, seed=4)
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    B, S, K, N = 2, 7, cfg.chunk_size, cfg.num_features
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, S, K, N)).astype(np.float32)
    dy = rng.standard_normal((B, S, K, N)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    y, tape = eng.train_path_forward(0, path, xt)
    y_inf = eng.stage_path(0, path, xt)
    assert O.agreement_db(y.cpu().numpy(), y_inf.cpu().numpy()) > 120
    dx = eng.train_path_backward(0, path, xt, torch.from_numpy(dy).to(dev), tape)
    torch.cuda.synchronize()

    ref = StockDPTN(cfg, sd)
    name = "intra_chunk_block" if path == 0 else "inter_chunk_block"
    pre = f"dprnn.model.0.{name}."
    _, _, rnn = ref.paths[path]
    rnn = rnn.double()
    params = {k: v.double().requires_grad_(True) for k, v in ref.sd.items() if k.startswith(pre)}
    ref.sd.update(params)
    for p in rnn.parameters():
        p.requires_grad_(True)
    xs = torch.from_numpy(x).double()
    seqs = (xs.reshape(B * S, K, N) if path == 0 else xs.transpose(1, 2).reshape(B * K, S, N)).requires_grad_(True)
    with torch.enable_grad():
        out = ref._path(seqs, pre, None, rnn)
        dys = torch.from_numpy(dy).double()
        out.backward(dys.reshape(B * S, K, N) if path == 0 else dys.transpose(1, 2).reshape(B * K, S, N))
    want_dx = seqs.grad.reshape(B, S, K, N) if path == 0 else seqs.grad.reshape(B, K, S, N).transpose(1, 2)
    assert O.agreement_db(dx.cpu().numpy(), want_dx.numpy()) > 80, "d x"
    want = {"rnn." + k: v.grad for k, v in rnn.named_parameters()}
    for leaf in ("fc.weight", "fc.bias", "norm1d.weight", "norm1d.bias"):
        want[leaf] = params[pre + leaf].grad
    for leaf, gref in want.items():
        got = grads[pre + leaf].cpu().numpy()
        assert got.shape == tuple(gref.shape), leaf
        assert O.agreement_db(got, gref.numpy()) > 70, (leaf, O.agreement_db(got, gref.numpy()))


@pytest.mark.parametrize("arch,features", [("dptn", 128), ("dptn", 64), ("dprnn", 64)])
def test_unidirectional_inter_path_backward_matches_autograd(dev, arch, features):
    """bidir = False (dptn.py:60, dprnn.py:56-63): the inter-chunk LSTM has ONE direction (FFN / fc over 128 hidden columns,
    BPTT and LSTM gradients for the forward direction only); block half with tape + backward against fp64 autograd."""
    from speech_separation_amd.engine import DptnEngine, params_to_device
    from speech_separation_amd.spec import DPRNN_AUDIO
    base = DPTN_AV if arch == "dptn" else DPRNN_AUDIO
    cfg = DPTNConfig(**{**base.to_dict(), "num_blocks": 1, "dropout": 0.0, "num_features": features, "hidden_video": features,
                        "audio_only": True, "bidir": False, "chunk_size": 50, "step_size": 25})
    sd = synthetic_state_dict(cfg, seed=4)
    assert f"dprnn.model.0.inter_chunk_block.rnn.weight_ih_l0_reverse" not in sd
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    grads = eng.bind_grads()
    B, S, K, N, path = 2, 9, cfg.chunk_size, cfg.num_features, 1
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, S, K, N)).astype(np.float32)
    dy = rng.standard_normal((B, S, K, N)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    y, tape = eng.train_path_forward(0, path, xt)
    y_inf = eng.stage_path(0, path, xt)
    assert O.agreement_db(y.cpu().numpy(), y_inf.cpu().numpy()) > 120
    dx = eng.train_path_backward(0, path, xt, torch.from_numpy(dy).to(dev), tape)
    torch.cuda.synchronize()
    ref = StockDPTN(cfg, sd)
    pre = "dprnn.model.0.inter_chunk_block."
    _, mha, rnn = ref.paths[path]
    rnn = rnn.double()
    if mha is not None:
        mha = mha.double().train(False)
    params = {k: v.double().requires_grad_(True) for k, v in ref.sd.items() if k.startswith(pre)}
    ref.sd.update(params)
    for p in (list(mha.parameters()) if mha is not None else []) + list(rnn.parameters()):
        p.requires_grad_(True)
    seqs = torch.from_numpy(x).double().transpose(1, 2).reshape(B * K, S, N).requires_grad_(True)
    with torch.enable_grad():
        out = ref._path(seqs, pre, mha, rnn)
        out.backward(torch.from_numpy(dy).double().transpose(1, 2).reshape(B * K, S, N))
    want_dx = seqs.grad.reshape(B, K, S, N).transpose(1, 2)
    assert O.agreement_db(dx.cpu().numpy(), want_dx.numpy()) > 80, "d x"
    want = {"rnn." + k: v.grad for k, v in rnn.named_parameters()}
    if mha is not None:
        want.update({"mha.in_proj_weight": mha.in_proj_weight.grad, "mha.in_proj_bias": mha.in_proj_bias.grad,
                     "mha.out_proj.weight": mha.out_proj.weight.grad, "mha.out_proj.bias": mha.out_proj.bias.grad})
    for leaf in (("ln1.weight", "ln1.bias", "ffn.1.weight", "ffn.1.bias", "ln2.weight", "ln2.bias") if arch == "dptn"
                 else ("fc.weight", "fc.bias", "norm1d.weight", "norm1d.bias")):
        want[leaf] = params[pre + leaf].grad
    assert len(want) == sum(k.startswith(pre) for k in grads)
    for leaf, gref in want.items():
        got = grads[pre + leaf].cpu().numpy()
        assert got.shape == tuple(gref.shape), leaf
        assert O.agreement_db(got, gref.numpy()) > 70, (leaf, O.agreement_db(got, gref.numpy()))


def test_unidirectional_whole_model_trains(dev):
    """A bidir = False DPTN-AV model through the drop-in module: loss.backward() fills every parameter's gradient (finite,
    non-zero) and three fused optimizer steps lower the loss."""
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.spec import synthetic_inputs
    from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step
    model = DPTNAVWavEncDec(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128, num_blocks=2,
                            chunk_size=150, step_size=75, dropout=0.0, num_heads=4, bidir=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=3).items()})
    model = model.to(dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    inp = synthetic_inputs(model.cfg, B=3, T=4000, Tv=50, seed=5)
    losses = []
    for _ in range(3):
        st = train_step(model, {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}, SiSNRWavLoss(), opt, 10.0)
        losses.append(float(st["loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    for k, p_ in model.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all() and float(p_.grad.abs().max()) > 0, k
