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
