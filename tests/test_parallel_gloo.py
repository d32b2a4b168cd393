"""world_size-2 gloo rehearsal of the multi-GPU path: batch shards, no data-path collective, metric sums and
the max-over-ranks timing reduction (the same DistEnv code runs over RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import dptn_oracle as O
from speech_separation_amd.parallel import DistEnv, shard_range
from speech_separation_amd.spec import DPTN_TINY, synthetic_inputs, synthetic_state_dict


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 16, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    env = DistEnv.from_environ(expected_world=world, backend="gloo", device="cpu")
    cfg = DPTN_TINY
    sd = synthetic_state_dict(cfg, seed=7)
    B = 5
    inp = synthetic_inputs(cfg, B=B, T=209, Tv=9, seed=11)
    lo, hi = shard_range(B, env.rank, env.world)
    mine = {k: v[lo:hi] for k, v in inp.items()}
    out = O.forward(cfg, sd, **mine)                          # the oracle stands in for the GPU forward here
    # per-item SI-SNR sums are additive over shards -> one SUM all-reduce at the end of an evaluation
    sums = [0.0, 0.0]
    for i in range(hi - lo):
        sums[0] += O.si_snr_db(out["s1_pred"][i:i + 1], mine["s1"][i:i + 1])
        sums[1] += 1.0
    env.barrier()
    tot = env.sum_over_ranks(sums)
    slow = env.max_over_ranks(float(rank + 1))
    q.put((rank, lo, hi, tot, slow, out["s1_pred"]))
    env.close()


def test_two_rank_sharded_evaluation_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = DPTN_TINY
    sd = synthetic_state_dict(cfg, seed=7)
    inp = synthetic_inputs(cfg, B=5, T=209, Tv=9, seed=11)
    full = O.forward(cfg, sd, **inp)
    # shards tile the batch and reproduce the unsharded outputs bit for bit (independent mixtures)
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]
    got = np.concatenate([r[5] for r in res], 0)
    assert np.array_equal(got, full["s1_pred"])
    want = sum(O.si_snr_db(full["s1_pred"][i:i + 1], inp["s1"][i:i + 1]) for i in range(5))
    for r in res:
        assert abs(r[3][0] - want) < 1e-9 and r[3][1] == 5.0 and r[4] == 2.0


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from speech_separation_amd.train import allreduce_gradients
    env = DistEnv.from_environ(expected_world=world, backend="gloo", device="cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) / 10 + rank          # each rank: its own batch shard
    model(x).pow(2).mean().backward()
    allreduce_gradients(model, env)                                              # ONE flat-bucket collective
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()]))   # numpy: pickled by value
    env.close()


def test_gradient_allreduce_is_the_mean_over_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    want = [torch.zeros_like(p) for p in model.parameters()]
    for rank in range(world):
        model.zero_grad()
        x = torch.arange(20, dtype=torch.float32).reshape(4, 5) / 10 + rank
        model(x).pow(2).mean().backward()
        for w, p in zip(want, model.parameters()):
            w += p.grad / world
    for rank in range(world):
        for got, w in zip(res[rank][1], want):
            assert np.allclose(got, w.numpy(), atol=1e-6)


# ---- eight ranks (the node BASELINE configs[3] / [4] name): nothing beyond world 2 had ever run ---------------
def _spawn(target, world, *args, timeout=240):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


class _FlatGradNet(torch.nn.Module):
    """Parameters whose .grad are views of ONE flat tensor, as the drop-in module hands them to autograd
    (model._SeparateFn.backward): allreduce_gradients then reduces that tensor in place."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Parameter(torch.randn(6, 5))
        self.b = torch.nn.Parameter(torch.randn(11))
        self._flat_grad = torch.zeros(41)

    def fill(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self._flat_grad.copy_(torch.randn(41, generator=g))
        self.a.grad = self._flat_grad[:30].view(6, 5)
        self.b.grad = self._flat_grad[30:]


def _flat_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from speech_separation_amd.train import allreduce_gradients
    env = DistEnv.from_environ(expected_world=world, backend="gloo", device="cpu")
    net = _FlatGradNet()
    net.fill(rank)
    how = allreduce_gradients(net, env)
    lo, hi = shard_range(37, env.rank, env.world)             # ragged: 37 items over 8 ranks
    spans = env.gather_over_ranks(float(lo)), env.gather_over_ranks(float(hi))
    q.put((rank, how, net._flat_grad.numpy().copy(), net.a.grad.numpy().copy(), spans, env.backend_world()))
    env.close()


def test_eight_ranks_flat_gradient_bucket_is_the_mean_and_shards_tile_a_ragged_count():
    world = 8
    res = _spawn(_flat_worker, world)
    want = torch.zeros(41)
    for r in range(world):
        want += torch.randn(41, generator=torch.Generator().manual_seed(100 + r)) / world
    for rank, how, flat, a_grad, (los, his), bw in res:
        assert how == "flat-in-place" and bw == world
        assert np.allclose(flat, want.numpy(), atol=1e-6)
        assert np.array_equal(a_grad, flat[:30].reshape(6, 5))          # the parameters' .grad still ARE the flat tensor
        assert los[0] == 0 and his[-1] == 37 and los[1:] == his[:-1]     # every rank sees the same tiling, no gap, no overlap
        assert max(h - l for l, h in zip(los, his)) - min(h - l for l, h in zip(los, his)) <= 1


class _CpuSisnr:
    """stand-in metric with the drop-in's call contract (value per batch = batch mean), computed by the oracle"""
    name = "si_snr"

    def __call__(self, s1_pred, s1, **batch):
        return float(np.mean([O.si_snr_db(s1_pred[i:i + 1].numpy(), s1[i:i + 1].numpy()) for i in range(s1.shape[0])]))


def _oracle_model(cfg, sd):
    def model(mix, s1_embedding, s2_embedding, **batch):
        out = O.forward(cfg, sd, mix=mix.numpy(), s1_embedding=s1_embedding.numpy(), s2_embedding=s2_embedding.numpy())
        return {k: torch.from_numpy(v) for k, v in out.items()}
    return model


def _infer_worker(rank, world, port, q, root, n, bs):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import pickle
    from speech_separation_amd.evaluate import run_inference
    env = DistEnv.from_environ(expected_world=world, backend="gloo", device="cpu")
    with open(os.path.join(root, "entries.pkl"), "rb") as f:
        entries = pickle.load(f)
    cfg = DPTN_TINY
    logs, stats = run_inference(_oracle_model(cfg, synthetic_state_dict(cfg, seed=7)), entries, bs, [_CpuSisnr()],
                                save_dir=os.path.join(root, "pred"), device="cpu", workers=2, env=env)
    q.put((rank, logs, stats["items"], stats["files"]))
    env.close()


def test_eight_rank_inference_pipeline_equals_single_process(tmp_path):
    """evaluate.run_inference over 8 gloo ranks on a ragged dataset (23 items in batches of 3: 8 batches, the last one of 2):
    ranks take whole batches, the metric is the mean over batches as MetricTracker forms it, every item's file is written
    exactly once, and the logs equal the single-process value on every rank."""
    import pickle
    from tests.dataset_fixture import make_dataset
    from speech_separation_amd.evaluate import run_inference
    cfg = DPTN_TINY
    n, bs, world = 23, 3, 8
    entries, _ = make_dataset(str(tmp_path), n, T=209, Tv=9, emb=cfg.video_emb_size, seed=5)
    with open(tmp_path / "entries.pkl", "wb") as f:
        pickle.dump(entries, f)
    res = _spawn(_infer_worker, world, str(tmp_path), n, bs)
    single, st = run_inference(_oracle_model(cfg, synthetic_state_dict(cfg, seed=7)), entries, bs, [_CpuSisnr()], save_dir=None,
                               device="cpu", workers=2)
    assert st["items"] == n
    assert sum(r[2] for r in res) == n and sum(r[3] for r in res) == n
    assert [r[2] for r in res] == [3, 3, 3, 3, 3, 3, 3, 2]                       # one batch per rank, the ragged one last
    for r in res:
        assert abs(r[1]["si_snr"] - single["si_snr"]) < 1e-9
    assert len(os.listdir(tmp_path / "pred")) == n
