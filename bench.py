#!/usr/bin/env python3
"""
bench.py -- TFR Mpoints/s of the CWT + STX + entropy hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic records resident in HBM:
Gabor CWT panel + Stockwell panel (complex coefficients written once) with the tfr_info reductions
(per-band / per-time power, max, total, entropy sums) fused into the producing kernels, and --
with more than one rank -- one RCCL gather of the reduced product to rank 0.
Default workload = BASELINE.json configs[1]: 1 channel per GPU, 2^20 samples @ 1 kHz, order 3, fp32.
A "point" is one complex TFR coefficient; points per step = 2 * channels * bands * n per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="after the warmup steps, keep stepping (untimed) until this much wall time has passed since "
                         "their start: the GPU leaves its idle clocks only after ~0.1 s of load (0 = off)")
    ap.add_argument("--channels", type=int, default=1, help="records per GPU (weak scaling)")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--order", type=float, default=3.0)
    ap.add_argument("--fs", type=float, default=1000.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--engine", default="auto", choices=["auto", "hipfft", "native"])
    return ap.parse_args()


def algorithmic_bytes(n_ch, n_b, n, length, real_bytes):
    """signal read + atom-spectrum bank read once + complex panel write (+ marginals out)."""
    s_r, s_c = real_bytes, 2 * real_bytes
    cwt = n_ch * n * s_r + n_b * length * s_c + n_ch * n_b * n * s_c + n_ch * (n_b + n) * s_r
    stx = n_ch * n * s_r + n_ch * n_b * n * s_c + n_ch * (n_b + n) * s_r
    return cwt, stx


def cpu_baseline(args, n, fs, order, budget_s):
    """The CPU oracle (NumPy restatement of the reference, oracle/tfr_oracle.py) timed on this host's
    cores on a bounded sample of the same workload: bands of the same record, CWT + STX + entropy
    sums per band, until the time budget is used."""
    from oracle import tfr_oracle as orc

    f = orc.band_table(fs, n, order)
    order_idx = list(range(0, len(f), 4)) + [j for j in range(len(f)) if j % 4]
    done = 0
    t0 = time.perf_counter()
    channel = 0
    while time.perf_counter() - t0 <= budget_s:  # whole records until the budget is used, then the bands that still fit
        x = orc.synth_chirp(n, fs, channel, channel + 1, np.float32 if args.dtype == "f32" else np.float64)
        for j in order_idx:
            _, _, c = orc.cwt_fft(order, x, fs, bands=[j])
            _, _, s = orc.stx_fft(order, x, fs, bands=[j])
            for panel in (c, s):
                p = np.abs(panel) ** 2
                _ = p.sum(), p.max(), np.sum(p * np.log2(p + orc.EPS64))
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        channel += 1
    dt = time.perf_counter() - t0
    cores = 1
    return {
        "value": round(2 * done * n / dt / 1e6, 3),
        "unit": "Mpoints/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} band-transform pairs ({done / len(f):.2f} records of {len(f)} bands, every 4th band first) of CWT+STX+entropy sums at n=2^{args.log2n}, "
                  f"order {order:g}, {dt:.1f} s of single-thread NumPy/SciPy pocketfft; host has "
                  f"{len(os.sched_getaffinity(0))} cores available",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import quantum_inferno_amd as qi
    from quantum_inferno_amd import _lib, dist as qdist, synth

    n, fs, order = 1 << args.log2n, args.fs, args.order
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    real_bytes = 4 if args.dtype == "f32" else 8
    n_ch = args.channels
    total_ch = n_ch * world
    first, _ = qdist.shard(total_ch, rank, world)
    bands = qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
    n_b = len(bands)
    engine_code = {"auto": _lib.QI_ENGINE_AUTO, "hipfft": _lib.QI_ENGINE_HIPFFT, "native": _lib.QI_ENGINE_NATIVE}[args.engine]
    plan = qi.TfrPlan(n, tdtype, dev, qi.TfrPlan.workspace_for(n, n_b, tdtype, n_ch, cap_bytes=64 << 30), engine_code)
    plan.set_styx_bank(order, fs)
    plan.set_stx_bands(order, fs)
    sig = torch.from_numpy(synth.channels(n, fs, first, n_ch, total_ch, np.float32 if tdtype == torch.float32 else np.float64)).to(dev)

    # the reduced products of both transforms live in one buffer: the message of the gather, no packing copy.  With more
    # than one rank the gather of step k overlaps the transforms of step k + 1 (two sets of outputs, used in turn).
    slots = qdist.reduced_slots(n_ch, n_b, n, tdtype)
    depth = 2 if world > 1 else 1
    messages = [torch.empty(2 * slots, dtype=torch.float64, device=dev) for _ in range(depth)]
    outs = [plan.cwt_stx(sig, coef=True, reductions=True, reduced_out=(m[:slots], m[slots:])) for m in messages]
    pipe = qdist.GatherPipeline(depth=depth, dst=0)

    def step():
        if world == 1:
            plan.cwt_stx(sig, out=outs[0])  # qi_cwt_stx: both transforms of the same records in one call
            return qdist.pack_reduced(list(outs[0]))
        i = pipe.acquire()
        plan.cwt_stx(sig, out=outs[i])
        return pipe.submit(i, qdist.pack_reduced(list(outs[i])))

    def fence():
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_warm = time.perf_counter()
    for _ in range(args.warmup):
        step()
    # a step is a third of a millisecond: a few warmup steps end long before the clocks have left idle
    settle_steps = 0
    while True:
        torch.cuda.synchronize()
        done = (time.perf_counter() - t_warm) * 1e3 >= args.settle_ms
        if world > 1:  # every rank runs the same number of steps (each step ends in a collective)
            flag = torch.tensor([1.0 if done else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item() > 0.5)
        if done:
            break
        for _ in range(10):
            step()
        settle_steps += 10
    # untimed: every stage timed with HIP events, to find the dominant stage and report the breakdown
    plan.profile(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stage_all = plan.profile_read()
    dominant = max(stage_all.items(), key=lambda kv: kv[1][0])[0]
    # timed region: HIP events around the dominant stage's launches only, on every 7th transform call (odd, so that the CWT and the
    # Stockwell calls of a step are sampled alike) -- every event is a bubble in the stream
    plan.profile(True, stages=[dominant], period=7)  # the fused call still ticks once per transform
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stage = plan.profile_read()
    plan.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        points_step = 2 * total_ch * n_b * n
        value = points_step * args.steps / dt / 1e6
        length = 2 * n
        alg_cwt, alg_stx = algorithmic_bytes(n_ch, n_b, n, length, real_bytes)
        # dominant kernel = the stage with the largest summed device time on this rank (native engine: pass 2, the
        # fused inverse-FFT row pass + epilogue; hipFFT engine: the batched inverse transform)
        name, (ms, launches) = dominant, stage[dominant]
        per_launch_ms = ms / max(launches, 1)
        launches_per_step = max(stage_all[dominant][1] / 3, 1)  # spans of that stage per step
        # algorithmic bytes of that stage: the complex coefficients of the bands it produces, written once (SURVEY s8d:
        # C*B*n*s_c), plus the per-time / per-band marginals it leaves behind
        sb = plan.stage_bands(name)
        stage_bands = sb[0] + sb[2]  # styx CWT + Stockwell panels of one step
        alg_kernel_step = n_ch * stage_bands * n * 2 * real_bytes + 2 * n_ch * n * real_bytes + n_ch * stage_bands * real_bytes
        alg_per_launch = alg_kernel_step / launches_per_step
        achieved = alg_per_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"{name}:{args.dtype}:n{args.log2n}:o{order:g}:c{n_ch}")
            except Exception:
                traffic = None
        dev_ms = sum(v[0] for v in stage_all.values()) / 3
        line = {
            "metric": "TFR Mpoints/sec (CWT+STX+entropy)",
            "value": round(value, 1),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: {n_ch} channel(s) per GPU x 2^{args.log2n} samples @ {fs:g} Hz, "
                            f"order N={order:g}, CWT+STX+entropy, {n_b} bands",
                "channels_per_gpu": n_ch,
                "n": n,
                "bands": n_b,
                "points_per_step": points_step,
                "engine": args.engine,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": name,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "launch_ms": round(per_launch_ms, 4),
                "algorithmic_bytes_per_launch": int(alg_per_launch),
                "bands_per_step": stage_bands,
            },
            "step_roofline": {
                "algorithmic_bytes_per_step": int(alg_cwt + alg_stx),
                "device_ms_per_step": round(dev_ms, 4),
                "achieved_gbs": round((alg_cwt + alg_stx) / (dev_ms * 1e-3) / 1e9, 1) if dev_ms > 0 else None,
                "frac": round((alg_cwt + alg_stx) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dev_ms > 0 else None,
                "stage_ms_per_step": {k: round(v[0] / 3, 4) for k, v in stage_all.items() if v[1]},
                "note": "stage breakdown from 3 untimed steps with every stage under HIP events",
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, n, fs, order, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
