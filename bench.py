#!/usr/bin/env python3
"""bench.py -- mixtures/sec of the DPTN-AV forward (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one forward of the whole hot path (libdptnav: head -> 12 TransformerDPRNN -> tail) over one batch of
16 synthetic 2-speaker mixtures (T = 32000 samples = 4 s @ 8 kHz) PER GPU, inputs already resident in HBM.
Mixtures are independent, so ranks shard by batch with no data-path collective (weak scaling); the only
collectives are the timing barrier and a MAX over ranks of the elapsed time.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline"      -- the DOMINANT kernel class = the one with the largest device time per step in this very run
                     (since round 2 the fused attention block, not the recurrence): algorithmic FLOPs per launch / its mean
                     device time, measured live with HIP events recorded on the launch stream (dptnav_profile_*);
                     `roofline.kernels` carries the same figures for every MFMA-bound class (attention block, LSTM
                     recurrence, pre-activation GEMM, ...), so the line is self-consistent with profiles/*kernel_stats.csv,
  "other_configs" -- short legs of BASELINE configs[1] (DPTN audio-only) and configs[4] (DPRNN-AV, B=32 x 8 s) and
  "train_step"    -- of configs[3], appended to the N=1 headline run (each in a try/except: a failing optional leg is
                     recorded as {"error": ...} and never costs the headline number).  With N > 1 ranks "train_step" is the
                     DATA-PARALLEL step of configs[3]: every rank runs it on its own 16 mixtures, one flat 17.8 MB gradient
                     all-reduce per step over RCCL, time = max over ranks, value = whole-job mixtures/s (a watchdog
                     prints the line without it if a collective never returns); "other_configs" / "latency_b1" are
                     per-GPU figures and stay with the N=1 line,
  "cpu_baseline"  -- oracle/torch_stock.py (stock PyTorch CPU operators = what the reference runs on CPU)
                     timed on this box's host cores on a bounded sample (rank 0, N=1 only),
  "kernels_ms_per_step" -- device ms per forward by kernel class (same HIP-event measurement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.parallel import DistEnv  # noqa: E402
from speech_separation_amd.spec import (DPRNN_AV, DPTN_AUDIO, DPTN_AV, synthetic_inputs,  # noqa: E402
                                        synthetic_state_dict)

DDP_LEG_LIMIT_S = 240          # N > 1: the optional data-parallel training leg is abandoned after this (the headline is printed anyway)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 256 FLOP/clk x 2.4 GHz
T_SAMPLES = 32000              # "4 s @ 8 kHz"
BATCH_PER_GPU = 16
#: --config: (model config, default batch per GPU, samples, BASELINE.json configs[] entry)
CONFIGS = {
    "dptn_av": (DPTN_AV, 16, 32000, "configs[2]: DPTN-AV (dptn_wav_av) forward, precomputed lip embeddings"),
    "dptn_audio": (DPTN_AUDIO, 16, 32000, "configs[1]: DPTN audio-only (dptn_wav) forward"),
    "dptn_av_train": (DPTN_AV, 16, 32000, "configs[3]: DPTN-AV training step (PIT SI-SNR loss + AdamW, clip 10), attention "
                                          "dropout 0.1"),
    "dprnn_av": (DPRNN_AV, 32, 128000, "configs[4]: DPRNN-AV long utterance (8 s @ 16 kHz): reference DPRNNEncDec "
                                       "backbone + this repo's AV fusion head"),
}


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


PMC_TABLES = {"dptn_av": os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"),
              "dptn_audio": os.path.join(ROOT, "profiles", "r05_pmc_traffic_dptn_audio.json"),
              "dprnn_av": os.path.join(ROOT, "profiles", "r05_pmc_traffic_dprnn_av.json")}
PMC_TABLE = PMC_TABLES["dptn_av"]
PMC_TABLE_TRAIN = os.path.join(ROOT, "profiles", "r05_train_pmc_traffic.json")   # tools/gpu_train_traffic.sh


def pmc_table(config: str, path: str = None):
    """The committed per-kernel HBM-traffic table (tools/pmc_table.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes over `bench.py --pmc-run`, gfx950 corrections applied; its header names the commit and the command it was taken
    at).  bench.py cannot read PMC counters itself, so `traffic` figures are FROM THIS FILE, for the configuration it was
    measured on only (anything else reports null)."""
    try:
        d = json.load(open(path or PMC_TABLES.get(config, PMC_TABLE)))
        return d if d.get("config") == config else None
    except (OSError, ValueError):
        return None


def table_state(tab) -> str:
    """Whether a committed PMC table was taken with the kernels of THIS tree (its csrc_digest against build.source_digest())."""
    from speech_separation_amd.build import source_digest
    if not tab:
        return "no table"
    have = tab.get("csrc_digest")
    if have is None:
        return f"STALE? table of commit {tab.get('commit', '?')} predates the source digest: kernels may have changed since"
    return "current (kernel sources unchanged since the table was taken)" if have == source_digest() else \
        f"STALE: taken at commit {tab.get('commit', '?')} with other kernel sources (digest {have} != {source_digest()})"


def log(msg: str):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, sd, seconds_budget: float = 25.0):
    """Reference CPU path (stock PyTorch operators, all host cores), bounded sample: max mixtures/s over
    B in {1, 4} (CPU throughput falls with B)."""
    from oracle.torch_stock import StockDPTN
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline on {cores} host threads")
    model = StockDPTN(cfg, sd)
    best, parts = 0.0, []
    t_start = time.perf_counter()
    for B, reps in ((1, 5), (4, 2)):
        inp = synthetic_inputs(cfg, B=B, T=T_SAMPLES, Tv=50, seed=123)
        t = {k: torch.from_numpy(v) for k, v in inp.items()}
        if parts and time.perf_counter() - t_start > 0.4 * seconds_budget:
            break  # not enough budget left for the larger batch
        model(**t)  # warm-up
        log(f"cpu_baseline B={B} warm-up done at {time.perf_counter() - t_start:.1f} s")
        done, t0 = 0, time.perf_counter()
        for _ in range(reps):
            model(**t)
            done += 1
            if time.perf_counter() - t_start > seconds_budget:
                break
        dt = time.perf_counter() - t0
        rate = B * done / dt
        parts.append(f"B={B}x{done}: {rate:.3f} mix/s")
        best = max(best, rate)
        if time.perf_counter() - t_start > seconds_budget:
            break
    # BASELINE configs[0]: the reference's own CPU-runnable case (Conv-TasNet, batch 4) -- an extra CPU datum
    from oracle import convtasnet_stock as CT
    csd = {k: torch.from_numpy(v) for k, v in CT.synthetic_convtasnet_weights(0).items()}
    cmix = torch.from_numpy(synthetic_inputs(cfg, B=4, T=T_SAMPLES, Tv=50, seed=5)["mix"])
    CT.forward(csd, cmix)
    t0 = time.perf_counter()
    for _ in range(3):
        CT.forward(csd, cmix)
    conv_rate = 4 * 3 / (time.perf_counter() - t0)
    return {"value": round(best, 4), "unit": "mixtures/sec", "cores": cores, "kind": "port",
            "configs0_convtasnet_cpu": {"value": round(conv_rate, 3), "unit": "mixtures/sec", "batch": 4,
                                        "note": "oracle/convtasnet_stock.py = src/model/convtasnet.py with stock PyTorch ops"},
            "sample": "oracle/torch_stock.py (stock PyTorch CPU ops, fp32, eval/no_grad), T=32000, 1 warm-up + "
                      + "; ".join(parts) + "; max over B reported, ONE sample per batch size (1.3-1.9 mixtures/s between runs of a day); B=16 not sampled (28.6 s per forward on 8 cores, and CPU throughput "
                      "falls with B: BASELINE.md section 2)"}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU (importing torch does not
    initialise HIP), so it starts the N ranks as fresh children through torch.distributed.run -- one process per GPU,
    rendezvous on 127.0.0.1 -- lets them inherit stdout/stderr (rank 0 prints the JSON line) and returns their exit
    code.  Never an exec of a process that holds the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"launching {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


PROGRESS = {"phase": "-", "step": -1}      # what the N > 1 watchdog reports when a collective never returns


def train_step_leg(cfg, dev, T, B=16, warmup=4, steps=12, env=None):
    """BASELINE configs[3] next to the headline: a few optimizer steps of the same model at batch 16 per GPU -- forward with
    tape, device PIT SI-SNR loss, HIP backward, device clip + AdamW, attention dropout 0.1 -- so that the driver-run line
    carries a training-step figure too (`--config dptn_av_train` is the full-length measurement).  With N > 1 ranks EVERY
    rank calls this (collectives inside): each step all-reduces the 17.8 MB flat gradient over RCCL (train.py), the time
    is the maximum over ranks and `value` the whole job's mixtures/s.  A rank that cannot set the leg up says so in a
    handshake BEFORE the first collective of a step, and all ranks skip the leg together."""
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step
    world = env.world if env is not None else 1
    ready, err = 1.0, None
    try:
        kw = {k: v for k, v in cfg.to_dict().items() if k not in ("audio_only", "arch")}
        model = DPTNAVWavEncDec(**kw)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
        model = model.to(dev).train()
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        seed = 123 + (env.rank if env is not None else 0)
        batch0 = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=seed).items()}
        crit = SiSNRWavLoss()
        PROGRESS.update(phase="train_step: local set-up step", step=0)
        train_step(model, dict(batch0), crit, opt, 10.0)          # one local step (no collective): allocations, first launches
        torch.cuda.synchronize(dev)
        if world > 1:
            # that step used rank-specific data: put every replica back on the SAME weights and a fresh optimizer state, or
            # the data-parallel steps below would average gradients of different models
            model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
            opt = FusedAdamW(model.parameters(), lr=1e-3)
            torch.cuda.synchronize(dev)
    except Exception as e:      # noqa: BLE001
        ready, err = 0.0, e
    if world > 1 and env.sum_over_ranks([ready])[0] < world:
        raise RuntimeError(f"training leg skipped on all ranks: set-up failed on at least one ({err!r} here)")
    if err is not None:
        raise err
    ekw = {"env": env} if world > 1 else {}
    for i in range(warmup):
        PROGRESS.update(phase="train_step: warm-up (gradient all-reduce inside)", step=i)
        train_step(model, dict(batch0), crit, opt, 10.0, **ekw)
    torch.cuda.synchronize(dev)
    if world > 1:
        PROGRESS.update(phase="train_step: barrier before the timed steps", step=-1)
        env.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        PROGRESS.update(phase="train_step: timed step (gradient all-reduce inside)", step=i)
        stats = train_step(model, dict(batch0), crit, opt, 10.0, **ekw)
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0
    if world > 1:
        PROGRESS.update(phase="train_step: barrier after the timed steps", step=-1)
        env.barrier()
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    per_rank_ms, allreduce = None, None
    if world > 1:
        dt = env.max_over_ranks(dt)
        per_rank_ms = [round(1e3 * t / steps, 3) for t in env.gather_over_ranks(mine)]
        # the collective's own device time: events around train.allreduce_gradients on the step's stream, after the timed
        # region (same buffers, same 17.8 MB); every rank takes part, rank 0 reports max / min over ranks
        from speech_separation_amd.train import allreduce_gradients
        PROGRESS.update(phase="train_step: timing the gradient all-reduce alone", step=-1)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record()
            how = allreduce_gradients(model, env)
            b.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        all_ms = env.gather_over_ranks(ms)
        nbytes = 4 * int(model._flat_grad.numel())
        allreduce = {"how": how, "bytes": nbytes, "device_ms_median_max_over_ranks": round(max(all_ms), 4),
                     "device_ms_median_min_over_ranks": round(min(all_ms), 4),
                     "bus_gb_per_s": round(2.0 * (world - 1) / world * nbytes / (max(all_ms) * 1e-3) / 1e9, 2),
                     "note": "all_reduce(SUM) + div on the flat gradient tensor, HIP events on the step's stream; bus bandwidth = "
                             "2 (n-1)/n x bytes / time (ring convention)"}
    dt /= steps
    flops = 3.0 * model._engine.flops_per_mixture(T) * B
    res = {"workload": f"configs[3]: DPTN-AV training step (PIT SI-SNR loss + AdamW, clip 10), batch={B} per GPU, T={T}, dropout 0.1",
           "value": round(world * B / dt, 3), "unit": "mixtures/sec", "n_gpus": world, "ms_per_step": round(1e3 * dt, 3), "steps": steps,
           "warmup": warmup, "frac": round(flops / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
           "parallelism": f"dp{world}" + (f" (one flat 17.8 MB gradient all-reduce per step over {'RCCL' if env.backend == 'nccl' else env.backend}, then clip + AdamW on every rank)" if world > 1 else ""),
           "note": "frac = 3 x forward FLOPs (612 GFLOP per mixture) / time / fp32-MFMA peak, per GPU; no host synchronisation in the step",
           "last_loss": round(float(stats["loss"]), 4)}
    if world > 1:
        res["per_rank_ms_per_step"] = per_rank_ms
        res["per_rank_spread_ms"] = round(max(per_rank_ms) - min(per_rank_ms), 3)
        res["gradient_allreduce"] = allreduce
    del model, opt, batch0
    torch.cuda.empty_cache()
    return res


def bench_train(args, env, cfg, B, T, workload):
    """BASELINE configs[3]: one optimizer step per bench step (zero_grad, forward with tape, device PIT SI-SNR loss, HIP
    backward, one flat-bucket RCCL all-reduce of the 17.8 MB of gradients when N > 1, device clip + AdamW); nothing in
    the step synchronises with the host (loss / gradient norm are read once, after the timed region)."""
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step
    dev = env.device
    kw = {k: v for k, v in cfg.to_dict().items() if k not in ("audio_only", "arch")}   # dropout 0.1 as in dptn_wav_av.yaml
    model = DPTNAVWavEncDec(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
    model = model.to(dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)     # dptn_wav_av.yaml:9-11 (AdamW, lr 1e-3, torch defaults)
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=123 + env.rank)
    batch0 = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    crit = SiSNRWavLoss()
    for kv in args.opt:                                # (experiments / A-B runs, tools/train_ab.py)
        model._get_engine(dev).set_option(kv.split("=")[0], int(kv.split("=")[1]))
    log(f"rank {env.rank}/{env.world} on {dev}: training warm-up")
    stats = None
    for _ in range(args.warmup):
        stats = train_step(model, dict(batch0), crit, opt, 10.0, env=env)
    env.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = train_step(model, dict(batch0), crit, opt, 10.0, env=env)
    torch.cuda.synchronize(dev)
    env.barrier()
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0
    elapsed = env.max_over_ranks(mine)
    per_rank_s = env.gather_over_ranks(mine)
    if args.pmc_run:
        if env.rank == 0:
            print(json.dumps({"pmc_run": True, "config": args.config, "steps": args.steps, "ms_per_step": round(1e3 * elapsed / args.steps, 3)}),
                  flush=True)
        env.close()
        return
    eng = model._engine
    eng.profile(True)
    eng.profile_reset()
    psteps = max(1, min(args.steps, 3))
    for _ in range(psteps):
        train_step(model, dict(batch0), crit, opt, 10.0, env=env)
    prof = eng.profile_read()
    eng.profile(False)
    if env.rank == 0:
        value = env.world * B * args.steps / elapsed
        ttab = pmc_table(args.config, PMC_TABLE_TRAIN)
        flops = 3.0 * eng.flops_per_mixture(T)            # forward + backward (dgrad + wgrad), recomputes not counted
        print(json.dumps({
            "metric": "mixtures/sec DPTN-AV training step (fwd + PIT SI-SNR loss + bwd + clip + AdamW)",
            "value": round(value, 3), "unit": "mixtures/sec", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}, batch={B} per GPU, T={T}, random-init weights (numpy seed 0)",
                       "batch_per_gpu": B, "samples": T,
                       "parallelism": f"dp{env.world} (one flat gradient all-reduce per step over RCCL)"},
            "roofline": {"bound": "mfma", "kernel": "whole step", "achieved": round(value / env.world * flops / 1e12, 3),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(value / env.world * flops / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": ttab["bytes_per_step"] if ttab and B == ttab.get("batch") and T == ttab.get("samples") else None,
                         "traffic_unit": (f"HBM bytes per optimizer step per GPU, from {os.path.relpath(PMC_TABLE_TRAIN, ROOT)} "
                                          f"(PMC FETCH_SIZE x2 + WRITE_SIZE summed over the step's kernels, commit {ttab.get('commit', '?')})")
                         if ttab else None,
                         "note": "algorithmic FLOPs = 3 x forward (612 GFLOP per mixture)"},
            "kernels_ms_per_step": {k: round(v[0] / psteps, 3) for k, v in prof.items()},
            "rccl_ranks": env.backend_world() if env.backend == "nccl" else 0, "backend": env.backend,
            "per_rank_mixtures_per_sec": [round(B * args.steps / t, 3) for t in per_rank_s],
            "last_step": {k: round(float(v), 5) for k, v in stats.items()} if stats else None}), flush=True)
    env.close()


def class_flops_per_step(cfg, prof, B, S, L):
    """Algorithmic FLOPs (2 per MAC, GEMM-like terms only: DESIGN.md section 4 = SURVEY.md 8d) of one forward step, by the
    kernel class that executes them IN THIS RUN: with the fused attention block (no qkv_gemm launches) the class
    `attention` carries in-projection + scores/PV + out-projection and the FFN of every path whose K6 rode in the next
    block's prologue (all but the ffn_ln_gemm launches that remain)."""
    N, H, K, nb = cfg.num_features, cfg.hidden_dim, cfg.chunk_size, cfg.num_blocks
    M = float(B) * S * K
    paths = [(K, 2), (S, 2 if cfg.bidir else 1)]          # (sequence length, LSTM directions) of the intra / inter path
    f = {"lstm_pre_gemm": sum(M * nb * nd * 2 * N * 4 * H for _, nd in paths),
         "lstm_recurrence": sum(M * nb * nd * 2 * H * 4 * H for _, nd in paths),
         "sep_gemm": M * 2 * N * 2 * N, "postproc_gemm": float(B) * L * (2 * 2 * N * N + 2 * 2 * N * cfg.kernel_size_enc)}
    ffn = sum(M * nb * 2 * nd * H * N for _, nd in paths)
    if prof["lstm_pre_gemm"][1] == 0:                      # input projection inside the recurrence (lstm16x.hip, 64 features)
        f["lstm_recurrence"] += f.pop("lstm_pre_gemm")
    if cfg.arch != "dptn":
        f["ffn_ln_gemm"] = ffn                             # DPRNN: fc + LayerNorm + residual (dprnn.py:40-46)
        return f, False
    qkv, outp = M * nb * 2 * 2 * N * 3 * N, M * nb * 2 * 2 * N * N
    att = sum(M * nb * 4 * ln * N for ln, _ in paths)
    n_att, n_ffn = prof["attention"][1], prof["ffn_ln_gemm"][1]
    fused = prof["qkv_gemm"][1] == 0 and n_att > 0
    if not fused:
        f.update(qkv_gemm=qkv, attention=att, outproj_ln_gemm=outp, ffn_ln_gemm=ffn)
        return f, False
    ride = 1.0 - n_ffn / float(n_att) if n_ffn < n_att else 0.0
    f["attention"] = qkv + att + outp + ffn * ride
    f["ffn_ln_gemm"] = ffn * (1.0 - ride)
    return f, ride > 0


def kernel_rows(cfg, eng, prof, psteps, B, T, dev):
    """One row per MFMA-bound kernel class: launches, mean launch time (HIP events), algorithmic FLOPs per launch, the
    fraction of the chip's fp32-MFMA peak; sorted by device time per step (row 0 = the dominant kernel)."""
    S, K = eng.chunks(T), cfg.chunk_size
    flops, ffn_rides = class_flops_per_step(cfg, prof, B, S, eng.frames(T))
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    ndir = 2 if cfg.bidir else 1
    fcln_on = eng.options_set.get("fcln", 1) != 0                    # (option fcln, default on: run_path / run_tail in dptnav.hip)
    dprnn_fc = fcln_on and cfg.arch != "dptn" and cfg.num_features == 64 and cfg.bidir
    fcln_ffn = fcln_on and cfg.arch == "dptn" and cfg.bidir
    rows = []
    for cls, fl in flops.items():
        ms, n = prof.get(cls, (0.0, 0))
        if n == 0 or fl <= 0:
            continue
        lps = n / float(psteps)
        tf = fl * psteps / (ms * 1e-3) / 1e12
        row = {"class": cls, "launches_per_step": lps, "launch_ms": round(ms / n, 4), "ms_per_step": round(ms / psteps, 4),
               "flops_per_launch": fl / lps, "achieved": round(tf, 3), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
        if cls == "lstm_recurrence":
            # which recurrence kernel those launches were (run_path in dptnav.hip: 16-sequence tiles when a launch fits
            # the chip in one round, i.e. for the DPTN half-batch launches; 32-sequence tiles otherwise).  A launch
            # occupies ONE CU per (direction, sequence tile) -- W_hh fills the CU's register file -- so beside `frac`
            # (against the whole chip's peak) the share of the chip the launch can use at all
            b_launch = B * 2 * cfg.num_blocks / lps
            fits16 = all(-(-int(b_launch * n_) // 16) * d <= n_cus for n_, d in ((S, 2), (K, ndir)))
            # ... and 4-sequence tiles (lstm4.hip, the low-latency kernel) while those are at most 1.15 rounds of the chip
            fits4 = fits16 and all(20 * (-(-int(b_launch * n_) // 4)) * d <= 23 * n_cus for n_, d in ((S, 2), (K, ndir)))
            tile = 4 if fits4 else (16 if fits16 else 32)
            wgs = sum(-(-int(b_launch * n_) // tile) * d for n_, d in ((S, 2), (K, ndir))) / 2.0     # mean of intra / inter
            occ = min(float(n_cus), wgs)
            kname = "lstm4_kernel" if fits4 else ("lstm16_kernel" if fits16 else "lstm_recurrence_kernel")
            if prof["lstm_pre_gemm"][1] == 0:
                tile, kname = 16, "lstm16x_kernel (x W_ih^T + b formed inside the recurrence: no K4 launch, no PRE tensor)"
                wgs = sum(-(-int(b_launch * n_) // tile) * d for n_, d in ((S, 2), (K, ndir))) / 2.0
                occ = min(float(n_cus), wgs)
            row.update(kernel=kname, pmc_match="lstm16x" if prof["lstm_pre_gemm"][1] == 0 else kname,
                       workgroups_per_launch=round(wgs, 1), cus_occupied=round(occ, 1), rounds=round(wgs / n_cus, 2),
                       frac_of_occupied_cus=round(tf / (PEAK_F32_MFMA_TFLOPS * occ / n_cus), 4),
                       algorithmic_bytes=int(B * S * K * 2 * cfg.num_blocks / lps) * ((cfg.num_features if prof["lstm_pre_gemm"][1] == 0 else 4 * cfg.hidden_dim * ndir) + cfg.hidden_dim * ndir) * 4)
        elif cls == "attention":
            fused = prof["qkv_gemm"][1] == 0
            row.update(kernel=("attn_block_kernel (in-proj + attention + out-proj + LN1" + (" + FFN/LN2 of the previous path)" if ffn_rides else ")"))
                       if fused else "attention_kernel (scores, softmax, PV)", pmc_match="attn_block_kernel<5, true>" if ffn_rides else
                       ("attn_block_kernel" if fused else "attention_kernel"))
        else:
            row.update(kernel={"lstm_pre_gemm": "gemm_ws_kernel<..., EpiLstmPre*> (x W_ih^T + b, fragment-order store)",
                               "ffn_ln_gemm": "fcln_kernel (DPRNN: LayerNorm(h W_fc^T + b) + x on 16-token tiles)" if dprnn_fc else
                               ("fcln_kernel (LayerNorm 2 of ReLU(h) W_f^T + b + y1 on 16-token tiles: the paths whose FFN does not ride in "
                                "the next attention block)" if fcln_ffn else "gemm_ws_kernel<..., EpiBiasResLN> (ReLU(h) W_f^T + b + y1, LayerNorm 2)"),
                               "qkv_gemm": "gemm_ws_kernel<..., EpiBiasStore> (in-projection)",
                               "outproj_ln_gemm": "gemm_ws_kernel<..., EpiBiasResLN> (out-projection + residual + LayerNorm 1)",
                               "sep_gemm": "fcln_kernel, plain form (PReLU + 1x1 conv N -> 2N on 16-token tiles)" if fcln_on else
                               "gemm_ws_kernel<..., ALoadDensePReLU, EpiBiasStore> (PReLU + 1x1 conv N -> 2N)",
                               "postproc_gemm": "taps_fold_kernel (OLA gather + post-processing conv + skip + decoder taps)"}[cls],
                       pmc_match={"lstm_pre_gemm": "EpiLstmPre", "postproc_gemm": "taps_fold_kernel",
                                  "sep_gemm": f"fcln_kernel<{cfg.num_features}, {2 * cfg.num_features}," if fcln_on else "ALoadDensePReLU",
                                  "ffn_ln_gemm": f"fcln_kernel<256, {cfg.num_features}," if (dprnn_fc or fcln_ffn) else
                                  "ALoadColsT<false>, EpiBiasResLN"}.get(cls))
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def profile_passes(eng, run_fn, psteps: int, psteps_alone: int):
    """Per-kernel device time by HIP events on the launch stream (dptnav_profile_*), twice over the same workload:
    as run (sub-batches on internal streams: up to three kernels share the chip, so a launch's duration is its SHARE of
    the CUs, not its speed) and SERIALISED (option serialize: the very same sub-batch launches, one after the other on one
    stream -- every launch alone on the chip; time the kernel actually owns)."""
    eng.profile(True)
    eng.profile_reset()
    for _ in range(psteps):
        run_fn()
    prof = eng.profile_read()
    eng.set_option("serialize", 1)
    try:
        eng.profile_reset()
        for _ in range(psteps_alone):
            run_fn()
        prof_alone = eng.profile_read()
    finally:
        eng.set_option("serialize", 0)
        eng.profile(False)
    return prof, prof_alone


def merged_rows(rows_run, rows_alone, table, B, T):
    """One row per MFMA-bound class: the serialised (alone-on-the-chip) figures are `launch_ms` / `achieved` / `frac`, the
    as-run ones sit under `as_run`; `traffic` from the committed PMC table when it covers this configuration and batch."""
    by = {r["class"]: r for r in rows_run}
    ok = table is not None and table.get("batch") == B and table.get("samples") == T
    out = []
    for r in rows_alone:
        q = dict(r)
        a = by.get(r["class"], {})
        q["as_run"] = {k: a.get(k) for k in ("launch_ms", "ms_per_step", "achieved", "frac")}
        krow = next((x for x in table["kernels"] if q.get("pmc_match") and q["pmc_match"] in x["name"]), None) if ok else None
        q["traffic"] = krow["hbm_bytes_per_launch"] if krow else None
        q.pop("pmc_match", None)
        out.append(q)
    return out, ok


def forward_leg(config: str, dev, steps: int, warmup: int, psteps: int, batch: int = 0):
    """A short, self-contained measurement of one forward configuration on this rank (N=1 legs of the headline line):
    mixtures/s, whole-path fraction of the fp32-MFMA peak and its dominant kernel's figures."""
    cfg, B_default, T, workload = CONFIGS[config]
    B = batch or B_default
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, seed=0), dev))
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=123)
    mix = torch.from_numpy(inp["mix"]).to(dev)
    e1 = torch.from_numpy(inp["s1_embedding"]).to(dev) if not cfg.audio_only else None
    e2 = torch.from_numpy(inp["s2_embedding"]).to(dev) if not cfg.audio_only else None
    out = (torch.empty_like(mix), torch.empty_like(mix))
    for _ in range(warmup):
        eng.forward(mix, e1, e2, out=out)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(mix, e1, e2, out=out)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    prof, prof_alone = profile_passes(eng, lambda: eng.forward(mix, e1, e2, out=out), psteps, psteps)
    tab = pmc_table(config)
    rows, tab_ok = merged_rows(kernel_rows(cfg, eng, prof, psteps, B, T, dev), kernel_rows(cfg, eng, prof_alone, psteps, B, T, dev),
                               tab, B, T)
    tf = B / dt * eng.flops_per_mixture(T) / 1e12
    res = {"workload": f"{workload}, batch={B}, T={T}", "value": round(B / dt, 3), "unit": "mixtures/sec",
           "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "gflop_per_mixture": round(eng.flops_per_mixture(T) / 1e9, 1),
           "whole_path_tflops": round(tf, 2), "whole_path_frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
           "dominant_kernel": {k: rows[0][k] for k in ("class", "kernel", "launches_per_step", "launch_ms", "ms_per_step", "achieved", "frac",
                                                       "as_run", "traffic") if k in rows[0]} if rows else None,
           "kernels": [{"class": r["class"], "launch_ms": r["launch_ms"], "ms_per_step": r["ms_per_step"], "frac": r["frac"],
                        "frac_as_run": r["as_run"]["frac"], "traffic": r["traffic"]} for r in rows],
           "kernels_note": "launch_ms / ms_per_step / frac: the launch alone on the chip (serialised pass); frac_as_run: beside the "
                           "other sub-batches' kernels",
           "traffic_table": (os.path.relpath(PMC_TABLES[config], ROOT) + ": " + table_state(tab)) if tab_ok else None,
           "outputs_finite": bool(torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all())}
    del eng, out, mix, e1, e2
    torch.cuda.empty_cache()
    return res


def latency_leg(dev, calls: int = 10):
    """The reference's OWN perf protocol (profiler.py:25-37,62-65: bs = 1, one 32000-sample mixture, 1 warm-up, 10 timed
    calls, mean / std of the call time) restated for this path: DPTN-AV alone (the lip-reading front end is outside the
    path: BASELINE configs[2] says precomputed embeddings), every call bracketed by a device synchronisation (the
    reference's loop has none, so its 0.0999 s on a P100 is a lower bound of its own latency)."""
    cfg, _, T, _ = CONFIGS["dptn_av"]
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, seed=0), dev))
    inp = synthetic_inputs(cfg, B=1, T=T, Tv=50, seed=123)
    mix, e1, e2 = (torch.from_numpy(inp[k]).to(dev) for k in ("mix", "s1_embedding", "s2_embedding"))
    out = (torch.empty_like(mix), torch.empty_like(mix))
    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(calls):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append(1e3 * (time.perf_counter() - t0))
        return ts
    times = timed(lambda: eng.forward(mix, e1, e2, out=out))
    res = {"what": "profiler.py protocol (bs = 1, T = 32000): one DPTN-AV forward, synchronised per call", "calls": calls,
           "mean_ms": round(float(np.mean(times)), 3), "std_ms": round(float(np.std(times)), 3), "min_ms": round(min(times), 3),
           "serial_lstm_steps": 6 * (cfg.chunk_size + eng.chunks(T)),
           "recurrence": "lstm4.hip (4-sequence tiles, v_mfma_f32_4x4x1_16B_f32) for launches of up to 1.15 rounds of the chip",
           "reference_published": "0.09989 s mean / 0.04486 s std on a Kaggle P100, lip-reader included, no device sync (README.md:116-117)"}
    # the same on 16-sequence tiles (option lstm4 = 0), and small batches (synchronised call time, mean of `calls`)
    eng.set_option("lstm4", 0)
    res["mean_ms_16_sequence_tiles"] = round(float(np.mean(timed(lambda: eng.forward(mix, e1, e2, out=out)))), 3)
    eng.set_option("lstm4", 1)
    small = {}
    for b in (2, 4, 8):
        inp_b = synthetic_inputs(cfg, B=b, T=T, Tv=50, seed=123)
        mb, e1b, e2b = (torch.from_numpy(inp_b[k]).to(dev) for k in ("mix", "s1_embedding", "s2_embedding"))
        ob = (torch.empty_like(mb), torch.empty_like(mb))
        small[f"B={b}"] = round(float(np.mean(timed(lambda: eng.forward(mb, e1b, e2b, out=ob)))), 3)
    res["small_batches_ms"] = small
    del eng, out
    torch.cuda.empty_cache()
    return res


def e2e_inference_leg(dev, forward_mixtures_per_s: float, items: int = 2048, batch: int = 16, T: int = 32000):
    """The reference's Inferencer loop end to end (src/trainer/inferencer.py:98-167 over src/datasets/base_dataset.py:56-135,
    188-205): a synthetic dataset in the reference's formats on local disk (per item three PCM16 WAVs of 4 s at 8 kHz and two
    zlib-compressed .npz lip embeddings), `evaluate.run_inference` = loader threads -> pinned batches + async H2D -> DPTN-AV forward
    -> SI-SNRi on the device -> (optionally) async D2H + one .pth per item.  Items/s with and without the prediction writer, the
    ratio to the bare forward of THIS run, and each host stage alone so that the one that caps the loop is named.  2 048 items (128
    batches): the loop's fixed cost -- first batch loaded, last batch's metric and files, 0.08-0.09 s -- is reported on its own
    (`fill_drain_and_host_s`); at 1 024 items it was 4.5 % of the run, an evaluation set of the reference's size has thousands."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.evaluate import run_inference
    from speech_separation_amd.io import load_item, save_predictions, write_synthetic_dataset
    from speech_separation_amd.metrics import SISNRiMetric
    cores = host_cores()
    workers = max(2, min(cores - 2, 14))
    root = tempfile.mkdtemp(prefix="dptnav_e2e_")
    try:
        t0 = time.perf_counter()
        entries, _ = write_synthetic_dataset(os.path.join(root, "data"), n=items, T=T)
        t_write = time.perf_counter() - t0
        model = DPTNAVWavEncDec(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128,
                                num_blocks=6, chunk_size=150, step_size=75, num_heads=4)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
        model = model.to(dev).eval()
        met = [SISNRiMetric(name="SISNRiMetric")]
        run_inference(model, entries[:2 * batch], batch, met, save_dir=os.path.join(root, "warm"), device=dev, workers=workers,
                      target_sr=8000)
        res = {"items": items, "batch": batch, "loader_threads": workers, "host_cores": cores, "dataset_write_s": round(t_write, 1),
               "per_item": "3 PCM16 WAVs (4 s @ 8 kHz) + 2 np.savez_compressed (512 x 50 f32); one .pth per item out",
               "forward_mixtures_per_s_this_run": round(forward_mixtures_per_s, 1)}
        for key, save in (("no_writer", None), ("with_writer", os.path.join(root, "out"))):
            best = None
            for _ in range(2):      # page cache warm on both passes (the dataset was just written); best of two
                logs, st = run_inference(model, entries, batch, met, save_dir=save, device=dev, workers=workers, target_sr=8000)
                best = st if best is None or st["items_per_s"] > best["items_per_s"] else best
            res[key] = {"items_per_s": round(best["items_per_s"], 1), "seconds": round(best["seconds"], 3),
                        "ratio_to_forward": round(best["items_per_s"] / forward_mixtures_per_s, 4), "files": best["files"],
                        # wall clock of the loop minus its forwards at the bare rate: loading the first batch, the last batch's
                        # metric + files, everything the host did not hide -- a fixed cost per run, not per item
                        "fill_drain_and_host_s": round(best["seconds"] - items / forward_mixtures_per_s, 3)}
            res["si_snri_db"] = round(float(logs["SISNRiMetric"]), 4)
        # host stages alone: what the loaders and the writer can do without the GPU in the loop
        with ThreadPoolExecutor(max_workers=workers) as pool:
            t0 = time.perf_counter()
            loaded = list(pool.map(lambda e: load_item(e, 8000), entries))
            res["loaders_alone_items_per_s"] = round(items / (time.perf_counter() - t0), 1)
        fake = {"s1_pred": torch.zeros(batch, T), "s2_pred": torch.zeros(batch, T), "s1": torch.zeros(batch, T), "s2": torch.zeros(batch, T)}
        t0 = time.perf_counter()
        nb = 8
        for b in range(nb):
            save_predictions({**fake, "audio_path": [f"w{b}_{i}.wav" for i in range(batch)]}, os.path.join(root, "wr"))
        res["writer_alone_items_per_s"] = round(nb * batch / (time.perf_counter() - t0), 1)
        del loaded
        caps = {"forward": forward_mixtures_per_s, "loaders": res["loaders_alone_items_per_s"], "writer (one thread)": res["writer_alone_items_per_s"]}
        res["slowest_stage_alone"] = min(caps, key=caps.get)
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)
        torch.cuda.empty_cache()


def optional_leg(name, fn, *a, **kw):
    """Optional legs never cost the headline: an exception becomes {"error": ...} under the leg's key."""
    try:
        t0 = time.perf_counter()
        res = fn(*a, **kw)
        log(f"{name}: done in {time.perf_counter() - t0:.1f} s")
        import gc
        gc.collect()
        torch.cuda.empty_cache()          # the next leg starts from an empty caching allocator
        return res
    except Exception as e:      # noqa: BLE001
        log(f"{name} FAILED: {e!r}")
        torch.cuda.empty_cache()
        return {"error": f"{type(e).__name__}: {e}"[:500]}


def split_experiment(eng, mix, e1, e2, out, B, steps, warmup, dev):
    """Opt-in split-precision experiment (never the headline): option split_bf16, same workload, reported under its own key
    with its agreement to the fp32 run."""
    ref1, ref2 = out[0].clone(), out[1].clone()
    eng.set_option("split_bf16", 1)
    try:
        o2 = (torch.empty_like(mix), torch.empty_like(mix))
        for _ in range(max(2, warmup)):
            eng.forward(mix, e1, e2, out=o2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(steps):
            eng.forward(mix, e1, e2, out=o2)
        torch.cuda.synchronize(dev)
        dts = (time.perf_counter() - t1) / steps
        eng.profile(True)
        eng.profile_reset()
        for _ in range(3):
            eng.forward(mix, e1, e2, out=o2)
        prof_s = eng.profile_read()
        eng.profile(False)
    finally:
        eng.set_option("split_bf16", 0)

    def agree(a, b):
        return float(10 * torch.log10(a.double().pow(2).sum() / (a.double() - b.double()).pow(2).sum().clamp_min(1e-300)))
    return {"what": "OPT-IN experiment, not the headline (option split_bf16): LSTM recurrence, pre-activation / FFN GEMMs and "
                    "the attention block on bf16 MFMAs with every operand split into bf16 hi + lo (hi*hi + hi*lo + lo*hi, fp32 "
                    "accumulation); head, tail, softmax, LayerNorm, cell update in fp32 as in the headline",
            "kernels_ms_per_step": {k: round(v[0] / 3, 3) for k, v in prof_s.items() if v[1]},
            "value": round(B / dts, 3), "unit": "mixtures/sec (this rank)", "ms_per_step": round(1e3 * dts, 4),
            "agreement_db_vs_f32_run": round(min(agree(ref1, o2[0]), agree(ref2, o2[1])), 1), "budget_db": 51.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="dptn_av",
                    help="dptn_av (default) is the configuration BASELINE.json's metric is quoted on")
    ap.add_argument("--batch", type=int, default=0, help="mixtures per GPU per step (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the short configs[3] training-step measurement "
                    "that the default N=1 headline run appends as `train_step`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end inference leg (dataset on local disk -> loaders -> forward "
                    "-> metric -> one .pth per item) that the default N=1 headline run appends as `e2e_inference`")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short configs[1] / configs[4] legs that the "
                    "default N=1 headline run appends as `other_configs`")
    ap.add_argument("--no-split", action="store_true", help="skip the opt-in split-precision experiment leg")
    ap.add_argument("--pmc-run", action="store_true", help="warm-up + timed steps only (no per-kernel event pass, no isolated "
                    "pass, no CPU leg): the command rocprofv3 --pmc / --kernel-trace passes are taken over")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="engine option(s) for the headline engine "
                    "(dptnav_set_option; experiments / A-B runs): the line then carries them under config.options")
    ap.add_argument("--serialize", action="store_true", help="with --pmc-run: option serialize = 1 (the step's sub-batch launches one "
                    "after the other on one stream), the pass bench.py's roofline figures are taken from")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    # BENCH_REHEARSAL=1: N ranks share cuda:0 and talk over gloo -- a rehearsal of the N > 1 code path on a one-GPU box
    # (its numbers mean nothing); the real launch is one rank per GPU over RCCL
    if os.environ.get("BENCH_REHEARSAL") == "1":
        env = DistEnv.from_environ(expected_world=args.gpus, backend="gloo", device="cuda:0")
    else:
        env = DistEnv.from_environ(expected_world=args.gpus)
    dev = env.device
    torch.cuda.set_device(dev)
    cfg, B_default, T, workload = CONFIGS[args.config]
    B, Tv = args.batch or B_default, 50
    headline = args.config == "dptn_av" and not args.pmc_run and env.world == 1 and not args.batch and not args.opt
    if args.config != "dptn_av" or args.pmc_run:
        args.no_cpu_baseline = True     # the CPU leg is only defined for the headline configuration
    ddp_leg = args.config == "dptn_av" and not args.pmc_run and env.world > 1    # N > 1: data-parallel training leg on all ranks
    if not headline:
        args.no_other_configs = True
        if not ddp_leg:
            args.no_train_step = True
    if args.config.endswith("_train"):
        return bench_train(args, env, cfg, B, T, workload)

    sd = synthetic_state_dict(cfg, seed=0)                       # random-init weights of the named architecture
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=123 + env.rank)   # each rank: its own shard of mixtures
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(sd, dev))
    mix = torch.from_numpy(inp["mix"]).to(dev)
    e1 = torch.from_numpy(inp["s1_embedding"]).to(dev) if not cfg.audio_only else None
    e2 = torch.from_numpy(inp["s2_embedding"]).to(dev) if not cfg.audio_only else None
    out = (torch.empty_like(mix), torch.empty_like(mix))
    for kv in args.opt:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.serialize:
        if not args.pmc_run:
            raise SystemExit("--serialize is a profiling aid: use it with --pmc-run")
        eng.set_option("serialize", 1)

    log(f"rank {env.rank}/{env.world} on {dev}: warm-up")
    for _ in range(args.warmup):
        eng.forward(mix, e1, e2, out=out)
    env.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward(mix, e1, e2, out=out)
    torch.cuda.synchronize(dev)
    env.barrier()
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0
    elapsed = env.max_over_ranks(mine)
    per_rank_s = env.gather_over_ranks(mine)
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    if args.pmc_run:
        if env.rank == 0:
            print(json.dumps({"pmc_run": True, "config": args.config, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * elapsed / args.steps, 4)}), flush=True)
        env.close()
        return 0

    # ---- per-kernel device time: HIP events on the launch stream, same workload, separate pass ----------
    psteps = max(1, min(args.steps, 10))
    psteps_alone = max(1, min(args.steps, 5))
    prof, prof_alone = profile_passes(eng, lambda: eng.forward(mix, e1, e2, out=out), psteps, psteps_alone)
    for _ in range(2):
        eng.forward(mix, e1, e2, out=out)          # `out` = the default (overlapped) path's result again
    finite = bool(torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all())

    split = None
    if not args.no_split:
        split = optional_leg("split_bf16_experiment", split_experiment, eng, mix, e1, e2, out, B, args.steps, args.warmup, dev)

    if env.rank == 0:
        S, K, H = eng.chunks(T), cfg.chunk_size, cfg.hidden_dim
        M = B * S * K
        # HBM traffic: from the committed PMC table (same configuration, batch and kernel only), never extrapolated
        tab = pmc_table(args.config)
        # rows: one per MFMA-bound class, sorted by the device time the class OWNS per step (serialised pass: the same
        # sub-batch launches alone on the chip); the as-run figures (launches of different sub-batches share the CUs) under `as_run`
        rows, tab_ok = merged_rows(kernel_rows(cfg, eng, prof, psteps, B, T, dev),
                                   kernel_rows(cfg, eng, prof_alone, psteps_alone, B, T, dev), tab, B, T)
        dom = rows[0]
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        kernels = {k: round(v[0] / psteps, 4) for k, v in prof.items()}
        kernels_alone = {k: round(v[0] / psteps_alone, 4) for k, v in prof_alone.items()}
        value = env.world * B * args.steps / elapsed
        whole = None
        tpath = os.path.relpath(PMC_TABLES.get(args.config, PMC_TABLE), ROOT)
        traffic_unit = (f"HBM bytes per launch, from {tpath} (PMC FETCH_SIZE x2 + WRITE_SIZE; {table_state(tab)})") if tab_ok \
            else f"no PMC table for this configuration / batch ({tpath})"
        if tab_ok:
            min_bytes = eng.min_bytes_per_mixture(T) * B
            whole = {"bytes_per_step": tab["bytes_per_step"], "bytes_per_mixture": round(tab["bytes_per_step"] / B),
                     "ideal_bytes_per_mixture": round(min_bytes / B), "ratio_to_ideal": round(tab["bytes_per_step"] / min_bytes, 2),
                     "avg_tb_per_s": round(tab["bytes_per_step"] / (elapsed / args.steps) / 1e12, 3),
                     "source": f"{tpath} (sum over kernels x launches per step; {table_state(tab)}; "
                               f"{tab.get('corrections', '')})"}
        line = {
            "metric": "mixtures/sec (2-spk, 4 s @ 8 kHz) DPTN-AV forward" if args.config == "dptn_av"
                      else f"mixtures/sec {args.config} forward",
            "value": round(value, 3), "unit": "mixtures/sec", "n_gpus": env.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}, batch={B} per GPU, T={T}, random-init weights (numpy seed 0)",
                       "batch_per_gpu": B, "samples": T, "tokens_per_mixture": S * K, "options": args.opt or None,
                       "parallelism": f"dp{env.world} (batch shards, no data-path collective)"},
            "roofline": {"bound": "mfma", "kernel": dom["kernel"], "class": dom["class"], "achieved": dom["achieved"],
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                         "traffic": dom["traffic"], "traffic_unit": traffic_unit,
                         "launch_ms": dom["launch_ms"], "flops_per_launch": dom["flops_per_launch"],
                         "launches_per_step": dom["launches_per_step"], "ms_per_step": dom["ms_per_step"],
                         "measured": "HIP events on the launch stream around every launch of the class, in a SERIALISED pass of the same "
                                     "workload (option serialize: the step's own sub-batch launches, one after the other on one stream, "
                                     "nothing else on the chip) -- the time the kernel owns.  rocprofv3 --kernel-trace --stats of "
                                     "`bench.py --pmc-run --serialize` gives the same average (profiles/)",
                         "as_run": dom["as_run"],
                         "as_run_note": "the same launches inside the overlapped step: sub-batches run on separate streams and up to 3 "
                                        "kernels share the CUs, so an as-run launch is longer than it is alone and its frac is a share "
                                        "of the chip, not an efficiency (launches x as-run time may exceed the step)",
                         "check": {"dominant_ms_per_step": dom["ms_per_step"], "step_ms": round(1e3 * elapsed / args.steps, 4),
                                   # a recurrence launch holds ONE CU per (direction, 16-sequence tile) -- 94-114 of 256 at this batch --
                                   # and the sub-batches' launches run side by side: what it owns of the chip is time x CU share
                                   "dominant_cu_share": round(dom.get("cus_occupied", n_cus) / n_cus, 4),
                                   "dominant_chip_ms_per_step": round(dom["ms_per_step"] * dom.get("cus_occupied", n_cus) / n_cus, 4),
                                   "owns_less_than_the_step": bool(dom["ms_per_step"] * dom.get("cus_occupied", n_cus) / n_cus
                                                                   <= 1e3 * elapsed / args.steps),
                                   "sum_of_classes_ms_per_step_serialised": round(sum(kernels_alone.values()), 3)},
                         "frac_of_occupied_cus": dom.get("frac_of_occupied_cus"),
                         "dominant_by": "largest device time per step among the kernel classes, serialised pass",
                         "kernels": rows,
                         "whole_path_tflops": round(value / env.world * eng.flops_per_mixture(T) / 1e12, 3),
                         "whole_path_frac": round(value / env.world * eng.flops_per_mixture(T) / 1e12
                                                  / PEAK_F32_MFMA_TFLOPS, 4)},
            "whole_path_traffic": whole,
            "split_bf16_experiment": split,
            "kernels_ms_per_step": kernels,
            "kernels_ms_per_step_serialised": kernels_alone,
            "outputs_finite": finite,
            "backend": env.backend, "rccl_ranks": env.backend_world() if env.backend == "nccl" else 0,
            "per_rank_mixtures_per_sec": [round(B * args.steps / t, 3) for t in per_rank_s],
        }
    del eng, out, mix, e1, e2
    torch.cuda.empty_cache()
    # N > 1: the training leg is data parallel -- every rank runs it (one RCCL all-reduce of the flat gradient per step)
    ddp_train = None
    if env.world > 1 and not args.no_train_step:
        # the headline is measured; a collective that never returns in this optional leg must not cost it: after
        # DDP_LEG_LIMIT_S rank 0 prints the line without the leg and every rank leaves
        def give_up():
            where = f"{PROGRESS['phase']}, step {PROGRESS['step']}"
            if env.rank == 0:
                line["train_step"] = {"error": f"data-parallel training leg did not finish within {DDP_LEG_LIMIT_S} s; abandoned "
                                               f"(rank 0 outstanding in: {where}); process exit code 3"}
                print(json.dumps(line), flush=True)
            log(f"rank {env.rank}: training leg abandoned after {DDP_LEG_LIMIT_S} s, outstanding in: {where}")
            os._exit(3)      # the headline is printed, but a collective that never returned is NOT a clean run
        guard = threading.Timer(DDP_LEG_LIMIT_S, give_up)
        guard.daemon = True
        guard.start()
        ddp_train = optional_leg("train_step", train_step_leg, cfg, dev, T, B=B, env=env)
        guard.cancel()
    if env.rank == 0:
        if env.world == 1 and not args.no_other_configs:      # per-GPU figures: the N = 1 line carries them
            line["other_configs"] = {
                "dptn_audio": optional_leg("other_configs.dptn_audio", forward_leg, "dptn_audio", dev, steps=10, warmup=3, psteps=3),
                "dprnn_av": optional_leg("other_configs.dprnn_av", forward_leg, "dprnn_av", dev, steps=3, warmup=1, psteps=1)}
            line["latency_b1"] = optional_leg("latency_b1", latency_leg, dev)
        if not args.no_train_step:
            line["train_step"] = ddp_train if env.world > 1 else optional_leg("train_step", train_step_leg, cfg, dev, T)
        # (behind the training leg: measured in round 5, the training step that FOLLOWS this leg in the same process runs 15 % slower
        #  -- 148 -> 127 mixtures/s -- whatever the leg leaves behind (loader threads' OpenMP teams, pinned host buffers, allocator
        #  segments); the leg itself is not affected by what precedes it)
        if env.world == 1 and not args.no_other_configs and not args.no_e2e:
            line["e2e_inference"] = optional_leg("e2e_inference", e2e_inference_leg, dev, value)
        if env.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = optional_leg("cpu_baseline", cpu_baseline, cfg, sd)
            if "value" in line["cpu_baseline"]:
                line["speedup_vs_cpu_baseline"] = round(value / max(line["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(line), flush=True)
    env.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
