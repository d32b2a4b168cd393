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
  "roofline"     -- dominant kernel (lstm_recurrence): algorithmic FLOPs per launch / its mean device time,
                    measured live with HIP events recorded on the launch stream (dptnav_profile_*),
  "cpu_baseline" -- oracle/torch_stock.py (stock PyTorch CPU operators = what the reference runs on CPU)
                    timed on this box's host cores on a bounded sample (rank 0, N=1 only),
  "kernels"      -- device ms per forward by kernel class (same HIP-event measurement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.parallel import DistEnv  # noqa: E402
from speech_separation_amd.spec import (DPRNN_AV, DPTN_AUDIO, DPTN_AV, synthetic_inputs,  # noqa: E402
                                        synthetic_state_dict)

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


PMC_TABLE = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
PMC_TABLE_TRAIN = os.path.join(ROOT, "profiles", "r02_train_pmc_traffic.json")   # tools/gpu_train_traffic.sh


def pmc_table(config: str, path: str = None):
    """The committed per-kernel HBM-traffic table (tools/pmc_table.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes over `bench.py --pmc-run`, gfx950 corrections applied; its header names the commit and the command it was taken
    at).  bench.py cannot read PMC counters itself, so `traffic` figures are FROM THIS FILE, for the configuration it was
    measured on only (anything else reports null)."""
    try:
        d = json.load(open(path or PMC_TABLE))
        return d if d.get("config") == config else None
    except (OSError, ValueError):
        return None


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
                      + "; ".join(parts) + "; max over B reported; B=16 not sampled (28.6 s per forward on 8 cores, and CPU throughput "
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


def train_step_leg(cfg, dev, T, B=16, warmup=4, steps=12):
    """BASELINE configs[3] next to the headline (N=1): a few optimizer steps of the same model at batch 16 -- forward with
    tape, device PIT SI-SNR loss, HIP backward, device clip + AdamW, attention dropout 0.1 -- so that the driver-run line
    carries a training-step figure too (`--config dptn_av_train` is the full-length measurement)."""
    from speech_separation_amd import DPTNAVWavEncDec
    from speech_separation_amd.train import FusedAdamW, SiSNRWavLoss, train_step
    kw = {k: v for k, v in cfg.to_dict().items() if k not in ("audio_only", "arch")}
    model = DPTNAVWavEncDec(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.cfg, seed=0).items()})
    model = model.to(dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    batch0 = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=123).items()}
    crit = SiSNRWavLoss()
    for _ in range(warmup):
        train_step(model, dict(batch0), crit, opt, 10.0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        stats = train_step(model, dict(batch0), crit, opt, 10.0)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    flops = 3.0 * model._engine.flops_per_mixture(T) * B
    return {"workload": f"configs[3]: DPTN-AV training step (PIT SI-SNR loss + AdamW, clip 10), batch={B}, T={T}, dropout 0.1",
            "value": round(B / dt, 3), "unit": "mixtures/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
            "frac": round(flops / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "note": "frac = 3 x forward FLOPs (612 GFLOP per mixture) / time / fp32-MFMA peak; no host synchronisation in the step",
            "last_loss": round(float(stats["loss"]), 4)}


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
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
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
            "last_step": {k: round(float(v), 5) for k, v in stats.items()} if stats else None,
            "host_syncs_per_step": 0}), flush=True)
    env.close()


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
    ap.add_argument("--pmc-run", action="store_true", help="warm-up + timed steps only (no per-kernel event pass, no isolated "
                    "pass, no CPU leg): the command rocprofv3 --pmc / --kernel-trace passes are taken over")
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
    if args.config != "dptn_av" or args.pmc_run:
        args.no_cpu_baseline = True     # the CPU leg is only defined for the headline configuration
    if args.config != "dptn_av" or args.pmc_run or env.world > 1:
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
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    if args.pmc_run:
        if env.rank == 0:
            print(json.dumps({"pmc_run": True, "config": args.config, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * elapsed / args.steps, 4)}), flush=True)
        env.close()
        return 0

    # ---- per-kernel device time: HIP events on the launch stream, same workload, separate pass ----------
    eng.profile(True)
    eng.profile_reset()
    psteps = max(1, min(args.steps, 10))
    for _ in range(psteps):
        eng.forward(mix, e1, e2, out=out)
    prof = eng.profile_read()
    # the dominant kernel once more with the two half-batches NOT overlapped (kernel alone on the chip)
    eng.set_option("overlap", 0)
    eng.profile_reset()
    try:
        for _ in range(3):
            eng.forward(mix, e1, e2, out=out)
        prof_iso = eng.profile_read()
    except RuntimeError as e:   # e.g. a whole-batch launch exceeds the 32-bit token index range (DPRNN, B=32)
        log(f"isolated pass skipped: {e}")
        prof_iso = None
    eng.set_option("overlap", 1)
    eng.profile(False)
    finite = bool(torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all())

    # ---- opt-in split-precision experiment (never the headline): the LSTM recurrence on bf16 MFMAs with hi/lo-split
    #      operands (option split_bf16), same workload, reported under its own key with its agreement to the fp32 run ----
    split = None
    if not args.pmc_run:
        ref1, ref2 = out[0].clone(), out[1].clone()
        eng.set_option("split_bf16", 1)
        o2 = (torch.empty_like(mix), torch.empty_like(mix))
        for _ in range(max(2, args.warmup)):
            eng.forward(mix, e1, e2, out=o2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.forward(mix, e1, e2, out=o2)
        torch.cuda.synchronize(dev)
        dts = (time.perf_counter() - t1) / args.steps
        eng.profile(True)
        eng.profile_reset()
        for _ in range(3):
            eng.forward(mix, e1, e2, out=o2)
        prof_s = eng.profile_read()
        eng.profile(False)
        eng.set_option("split_bf16", 0)

        def agree(a, b):
            return float(10 * torch.log10(a.double().pow(2).sum() / (a.double() - b.double()).pow(2).sum().clamp_min(1e-300)))
        split = {"what": "OPT-IN experiment, not the headline (option split_bf16): LSTM recurrence, pre-activation / FFN GEMMs and "
                         "the attention block on bf16 MFMAs with every operand split into bf16 hi + lo (hi*hi + hi*lo + lo*hi, fp32 "
                         "accumulation); head, tail, softmax, LayerNorm, cell update in fp32 as in the headline",
                 "kernels_ms_per_step": {k: round(v[0] / 3, 3) for k, v in prof_s.items() if v[1]},
                 "value": round(B / dts, 3), "unit": "mixtures/sec (this rank)", "ms_per_step": round(1e3 * dts, 4),
                 "agreement_db_vs_f32_run": round(min(agree(ref1, o2[0]), agree(ref2, o2[1])), 1),
                 "budget_db": 51.0}
        del ref1, ref2, o2

    if env.rank == 0:
        S, K, H = eng.chunks(T), cfg.chunk_size, cfg.hidden_dim
        M = B * S * K
        # as run in the timed region: the batch is two overlapped halves, so one launch covers B/2 mixtures and shares
        # the chip with the other half's GEMM / attention kernels
        ms, n = prof["lstm_recurrence"]
        lstm_ms = ms / max(n, 1)
        launches_per_step = n / psteps
        lstm_flops = float(M) * 2 * (2 * H * 4 * H) * (2 * cfg.num_blocks) / launches_per_step   # per launch: both directions, h W_hh^T
        achieved = lstm_flops / (lstm_ms * 1e-3) / 1e12
        iso = None
        if prof_iso is not None:
            ms_i, n_i = prof_iso["lstm_recurrence"]
            iso_ms = ms_i / max(n_i, 1)
            iso_tflops = float(M) * 2 * (2 * H * 4 * H) / (iso_ms * 1e-3) / 1e12
            iso = {"note": "same kernel, whole batch in one launch, nothing else on the chip (option overlap=0)",
                   "launch_ms": round(iso_ms, 4), "achieved": round(iso_tflops, 3),
                   "frac": round(iso_tflops / PEAK_F32_MFMA_TFLOPS, 4)}
        kernels = {k: round(v[0] / psteps, 4) for k, v in prof.items()}
        # which recurrence kernel those launches were (run_path in dptnav.hip: 16-sequence tiles when they fit the
        # chip in one round, i.e. for half-batch launches; 32-sequence tiles otherwise)
        b_launch = B * 2 * cfg.num_blocks / launches_per_step
        ndir = 2 if cfg.bidir else 1
        fits16 = all(-(-int(b_launch * n) // 16) * d <= torch.cuda.get_device_properties(dev).multi_processor_count
                     for n, d in ((S, 2), (K, ndir)))
        lstm_kernel = "lstm16_kernel" if fits16 else "lstm_recurrence_kernel"
        # a recurrence launch occupies ONE CU per (direction, sequence tile) -- W_hh fills the CU's register file -- so
        # beside `frac` (against the whole chip's peak, as the contract defines it) the share of the chip it can use
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        tile = 16 if fits16 else 32
        wgs = sum(-(-int(b_launch * n) // tile) * d for n, d in ((S, 2), (K, ndir))) / 2.0    # mean of intra / inter
        cu_share = min(1.0, wgs / n_cus)
        if iso is not None:
            iso["kernel"] = "lstm_recurrence_kernel"
        # the other large kernel class, same convention (algorithmic FLOPs of one launch / its mean time as run): the
        # fused attention block, which since round 2 also carries the FFN half of the previous path
        other = None
        if args.config == "dptn_av" and prof["attention"][1]:
            a_ms, a_n = prof["attention"]
            N_ = cfg.num_features
            per_tok = 2 * N_ * 3 * N_ + 2 * N_ * N_ + 4 * ((K + S) / 2.0) * N_                  # QKV + out-proj + scores / PV
            fused_ffn = prof["ffn_ln_gemm"][1] < a_n                                             # K6 rides in the block
            per_tok += (2 * 2 * H * N_) * (1.0 - prof["ffn_ln_gemm"][1] / a_n) if fused_ffn else 0.0
            a_flops = float(M) * (2 * cfg.num_blocks) * per_tok / (a_n / psteps)
            a_tf = a_flops / (a_ms / a_n * 1e-3) / 1e12
            other = {"kernel": "attn_block_kernel (in-proj + attention + out-proj + LN1" + (" + FFN/LN2 of the previous path)" if fused_ffn else ")"),
                     "launch_ms": round(a_ms / a_n, 4), "launches_per_step": a_n / psteps, "achieved": round(a_tf, 3),
                     "frac": round(a_tf / PEAK_F32_MFMA_TFLOPS, 4),
                     "note": "as run, beside the other sub-batch's recurrence (146 of 256 CUs)"}
        value = env.world * B * args.steps / elapsed
        # HBM traffic: from the committed PMC table (same configuration, batch and kernel only), never extrapolated
        tab = pmc_table(args.config)
        traffic, traffic_unit, whole = None, "no PMC table for this configuration / kernel (profiles/r02_pmc_traffic.json)", None
        if tab is not None and tab.get("batch") == B and tab.get("samples") == T:
            krow = next((r for r in tab["kernels"] if r["name"].startswith(lstm_kernel)), None)
            if krow is not None:
                traffic = krow["hbm_bytes_per_launch"]
                traffic_unit = (f"HBM bytes per launch, from {os.path.relpath(PMC_TABLE, ROOT)} (PMC FETCH_SIZE x2 + WRITE_SIZE, "
                                f"commit {tab.get('commit', '?')})")
            min_bytes = eng.min_bytes_per_mixture(T) * B
            whole = {"bytes_per_step": tab["bytes_per_step"], "bytes_per_mixture": round(tab["bytes_per_step"] / B),
                     "ideal_bytes_per_mixture": round(min_bytes / B), "ratio_to_ideal": round(tab["bytes_per_step"] / min_bytes, 2),
                     "avg_tb_per_s": round(tab["bytes_per_step"] / (elapsed / args.steps) / 1e12, 3),
                     "source": f"{os.path.relpath(PMC_TABLE, ROOT)} (sum over kernels x launches per step, commit {tab.get('commit', '?')})"}
        line = {
            "metric": "mixtures/sec (2-spk, 4 s @ 8 kHz) DPTN-AV forward" if args.config == "dptn_av"
                      else f"mixtures/sec {args.config} forward",
            "value": round(value, 3), "unit": "mixtures/sec", "n_gpus": env.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}, batch={B} per GPU, T={T}, random-init weights (numpy seed 0)",
                       "batch_per_gpu": B, "samples": T, "tokens_per_mixture": S * K,
                       "parallelism": f"dp{env.world} (batch shards, no data-path collective)"},
            "roofline": {"bound": "mfma", "kernel": lstm_kernel, "achieved": round(achieved, 3),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic, "traffic_unit": traffic_unit,
                         "algorithmic_bytes": int(M * 2 * cfg.num_blocks / launches_per_step) * (8 * H + 2 * H) * 4,
                         "launch_ms": round(lstm_ms, 4), "flops_per_launch": lstm_flops,
                         "cus_occupied": round(wgs, 1), "frac_of_occupied_cus": round(achieved / (PEAK_F32_MFMA_TFLOPS * cu_share), 4),
                         "launches_per_step": launches_per_step,
                         "isolated": iso, "second_kernel": other,
                         "whole_path_tflops": round(value / env.world * eng.flops_per_mixture(T) / 1e12, 3),
                         "whole_path_frac": round(value / env.world * eng.flops_per_mixture(T) / 1e12
                                                  / PEAK_F32_MFMA_TFLOPS, 4)},
            "whole_path_traffic": whole,
            "split_bf16_experiment": split,
            "kernels_ms_per_step": kernels,
            "outputs_finite": finite,
        }
        if not args.no_train_step:
            del eng, out
            torch.cuda.empty_cache()
            line["train_step"] = train_step_leg(cfg, dev, T)
        if env.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd)
            line["speedup_vs_cpu_baseline"] = round(value / max(line["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(line), flush=True)
    env.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
