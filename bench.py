#!/usr/bin/env python3
"""
bench.py -- TFR Mpoints/s of the CWT + STX + entropy hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic records resident in HBM:
Gabor CWT panel + Stockwell panel (complex coefficients written once) with the tfr_info reductions
(per-band / per-time power, max, total, entropy sums) fused into the producing kernels, and --
with more than one rank -- one RCCL gather of the reduced product to rank 0.

    --config 1 (default)  BASELINE.json configs[1]: 1 channel per GPU, 2^20 samples @ 1 kHz, order 3, fp32
    --config 2            BASELINE.json configs[2]: 64 channels x 2^20 samples, order 12, the full stack: the
                          order-12 STFT (styx_fft.stft_from_sig) of every record is part of the step
    --channels / --order / --log2n / --stft override single items of the chosen config.

A "point" is one complex TFR coefficient of the CWT or Stockwell panel; points per step = 2 * channels * bands * n
per GPU (the STFT's coefficients are reported separately, they are < 1 % of the step's bytes).

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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
CONFIGS = {1: dict(channels=1, order=3.0, stft=0), 2: dict(channels=64, order=12.0, stft=1),
           # configs[4]: 24 h of 800 Hz infrasound (69 120 000 samples) x 1024 channels, chunks of 2^20 with a hop of 2^19, fp64,
           # streamed from the host.  The bench streams a bounded sample per GPU (a block of 16 of the 128 channels a GPU
           # would own, `--stream-chunks` of their 131 chunks) and reports the rate; steps = items.
           4: dict(channels=16, order=12.0, stft=0, dtype="f64", fs=800.0, stream=1)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="index into BASELINE.json configs; default: configs[1] as `value` plus the configs[2] and float64 legs "
                         "under their own keys (one rank), configs[3] = 64 records per GPU (N ranks)")
    ap.add_argument("--legs", default="", help="comma list of the default run's legs to keep (main, configs2, f64, configs4, configs1_per_gpu)")
    ap.add_argument("--settle-ms", type=float, default=3000.0,
                    help="after the warmup steps, keep stepping (untimed) until this much wall time has passed since "
                         "their start: the GPU leaves its idle clocks only after ~0.1 s of load, and a GPU phase of a few "
                         "seconds can be seen by an outside utilisation sampler (0 = off)")
    ap.add_argument("--channels", type=int, default=None, help="records per GPU (weak scaling)")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--order", type=float, default=None)
    ap.add_argument("--stft", type=int, default=None, help="1: the order-N STFT of every record is part of the step")
    ap.add_argument("--fs", type=float, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--stream", type=int, default=None, help="1: records streamed from the host in overlapped chunks (config 4)")
    ap.add_argument("--stream-chunks", type=int, default=35, help="chunks of the streaming sample (a 24 h record has 131)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the all-cores CPU leg (0: min(16, cores))")
    ap.add_argument("--engine", default="auto", choices=["auto", "hipfft", "native"])
    ap.add_argument("--workspace-gib", type=float, default=0.0, help="plan scratch (0: sized from the batch, <= 48 GiB)")
    ap.add_argument("--two-streams", type=int, default=1,
                    help="1: after the timed region also report the steps issued alternately on two streams (calls of one or two records, N = 1)")
    ap.add_argument("--wrappers", type=int, default=1,
                    help="1: also time the reference-signature wrappers NumPy in -> NumPy out (config 1 shape, one GPU)")
    ap.add_argument("--stub", type=int, default=0,
                    help="1: CPU rehearsal of the multi-rank plumbing (gloo, no GPU): the transforms are replaced by a stub that "
                         "writes rank- and step-dependent reduced products; everything else -- sharding, message buffers, the "
                         "pipelined gather, the barriers and the max-over-ranks timing, the JSON line -- is the real code")
    ap.add_argument("--stub-ms", type=float, default=0.0,
                    help="with --stub: every stub transform call takes this long (a fixed step time, so that a rehearsed N-rank "
                         "line can be checked against N x the one-rank line)")
    ap.add_argument("--force-collective", type=int, default=0,
                    help="1: a ONE-rank run initialises the process group too (nccl = RCCL, world size 1) and every step ends in "
                         "the real pipelined gather-to-self: the multi-rank code path on a one-GPU box")
    ap.add_argument("--stub-dump", default="", help="with --stub: rank 0 saves the last gathered buffers here (torch.save)")
    return ap.parse_args()


def required_bytes(n_ch, n_b, n, real_bytes):
    """Bytes one transform of the step HAS to move: records in, complex panel out, marginals out."""
    s_r, s_c = real_bytes, 2 * real_bytes
    return n_ch * n * s_r + n_ch * n_b * n * s_c + n_ch * (n_b + n) * s_r


def survey_bytes(n_ch, n_b, n, length, real_bytes):
    """SURVEY.md s8(d) as written: the CWT also reads an atom-spectrum bank of B x 2n complex values once.  The native
    engine never reads such a bank (its filters are evaluated in registers / from compact windows); kept as a labelled
    second number only."""
    return 2 * required_bytes(n_ch, n_b, n, real_bytes) + n_b * length * 2 * real_bytes


FP64_VALU_PEAK_GINST = 1024 * 2.4 / 4  # wave instructions / ns: 1024 SIMDs x 2.4 GHz, a double-precision wave64 instruction
# issues over 4 cycles (78.6 TFLOP/s of float64 vector FMA = 614.4 G wave-FMAs / s x 64 lanes x 2)


def load_traffic():
    """profiles/traffic.json: per-launch HBM bytes and per-item instruction counts from the builder's rocprofv3 --pmc passes of
    the same commands (kept under profiles/; not collected in the run that prints them)."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            return json.load(open(tfile))
        except Exception:
            pass
    return {}


# ---- CPU baseline (the oracle port), before anything touches the GPU: plain forked workers ----------------------------
def _cpu_pairs(order, fs, n, dtype_name, worker, n_workers, deadline):
    """Band-transform pairs (CWT + STX + entropy sums of one band of one record) until the deadline; worker w takes the
    bands j = w (mod n_workers) of record 0, then of record 1, ..."""
    from oracle import tfr_oracle as orc

    f = orc.band_table(fs, n, order)
    dt = np.float32 if dtype_name == "f32" else np.float64
    done, channel = 0, 0
    while time.perf_counter() < deadline:
        x = orc.synth_chirp(n, fs, channel, channel + 1, dt)
        for j in range(worker, len(f), n_workers):
            _, _, c = orc.cwt_fft(order, x, fs, bands=[j])
            _, _, s = orc.stx_fft(order, x, fs, bands=[j])
            for panel in (c, s):
                p = np.abs(panel) ** 2
                _ = p.sum(), p.max(), np.sum(p * np.log2(p + orc.EPS64))
            done += 1
            if time.perf_counter() >= deadline:
                break
        channel += 1
    return done


def _cpu_worker(args):
    return _cpu_pairs(*args)


def cpu_baseline(args, n, fs, order):
    """The CPU oracle (NumPy restatement of the reference, oracle/tfr_oracle.py) timed on this host's cores on a
    bounded sample of the same workload: one core first (a third of the budget), then one process per core of the
    box's CPU share (bands of the same records dealt round-robin)."""
    import multiprocessing as mp

    cores_avail = len(os.sched_getaffinity(0))
    workers = args.cpu_workers or min(16, cores_avail)
    t0 = time.perf_counter()
    one = _cpu_pairs(order, fs, n, args.dtype, 0, 1, t0 + args.cpu_seconds / 3)
    dt_one = time.perf_counter() - t0
    pool_budget = args.cpu_seconds * 2 / 3
    ctx = mp.get_context("fork")  # nothing has initialised the GPU yet (bench.py runs this leg first)
    with ctx.Pool(workers) as pool:
        pool.map(_cpu_worker, [(order, fs, 4096, args.dtype, w, workers, time.perf_counter() + 0.2) for w in range(workers)])
        t1 = time.perf_counter()
        counts = pool.map(_cpu_worker, [(order, fs, n, args.dtype, w, workers, t1 + pool_budget) for w in range(workers)])
        dt_all = time.perf_counter() - t1
    return {
        "value": round(2 * sum(counts) * n / dt_all / 1e6, 3),
        "unit": "Mpoints/s",
        "cores": workers,
        "kind": "port",
        "single_core_value": round(2 * one * n / dt_one / 1e6, 3),
        "sample": f"{sum(counts)} band-transform pairs (CWT + STX + entropy sums of one band of one 2^{args.log2n}-sample record, "
                  f"order {order:g}) in {dt_all:.1f} s on {workers} processes (one per core of the box's share; the host shows "
                  f"{cores_avail} cores), after {one} pairs in {dt_one:.1f} s on one core; single-thread NumPy/SciPy pocketfft each",
    }


class OwnedRecords:
    """[channels, samples] record set of which this process only materialises the rows it will stream (the 24 h job is
    566 GB: no rank holds it all).  Rows outside [first, first + count) raise -- a rank that touched one would have taken
    an item that is not its own."""

    def __init__(self, shape, first, rows):
        self.shape, self.ndim, self.dtype = tuple(shape), 2, rows.dtype
        self.first, self.rows = first, rows

    def __getitem__(self, key):
        ch, tm = key
        lo, hi = ch.start - self.first, ch.stop - self.first
        if lo < 0 or hi > self.rows.shape[0]:
            raise IndexError(f"channels {ch.start}:{ch.stop} are not owned by this rank ({self.first}:{self.first + self.rows.shape[0]})")
        return self.rows[lo:hi, tm]


def stream_bench(args, ctx, cpu):
    """BASELINE configs[4] as a bounded sample: float64 records on the host, overlapped chunks, the double-buffered
    pipeline of quantum_inferno_amd.stream (pinned staging + copy stream), reduced products only.  The record set is
    `world` blocks of `channels` records; its (block, chunk) items are dealt to the ranks by `stream.rank_items` -- the
    24 h job's sharding -- so every rank streams `chunks` items.  One step = one item: CWT + STX + entropy of `channels`
    records of 2^20 samples."""
    import torch
    import torch.distributed as dist

    import quantum_inferno_amd as qi
    from quantum_inferno_amd import stream, synth

    world, rank, dev, stub = ctx.world, ctx.rank, ctx.dev, ctx.stub
    n, fs, order = 1 << args.log2n, args.fs, args.order
    hop = n // 2
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    npd = np.float32 if args.dtype == "f32" else np.float64
    real_bytes = 4 if args.dtype == "f32" else 8
    n_ch = args.channels
    n_b = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
    warm_req = 3 if args.warmup is None else args.warmup
    chunks = max(args.stream_chunks, warm_req + 2)
    total = n + (chunks - 1) * hop
    # this rank's block of the record set (weak scaling: the 24 h job has 128 channels x 131 chunks per GPU)
    rows = np.empty((n_ch, total), dtype=npd)
    base = synth.log_chirp(n, fs, rank, max(world, 1), npd)
    rng = np.random.default_rng(1000 + rank)
    reps = -(-total // n)
    noise = 0.01 * rng.standard_normal(total + 7919 * n_ch).astype(npd)
    for c in range(n_ch):
        rows[c] = np.tile(np.roll(base, 7919 * c), reps)[:total] + noise[7919 * c : 7919 * c + total]
    host = OwnedRecords((n_ch * world, total), n_ch * rank, rows)
    if stub:
        plan = StubPlan(n, n_b, rank, dtype=tdtype, step_ms=args.stub_ms)
    else:
        plan = qi.TfrPlan(n, tdtype, dev, qi.TfrPlan.workspace_for(n, n_b, tdtype, n_ch, cap_bytes=32 << 30))
        plan.set_styx_bank(order, fs)
        plan.set_stx_bands(order, fs)
    pipe = stream.StreamPipeline(plan, host, hop, block=n_ch, transforms=("cwt", "stx"), keep_time=False)
    mine = stream.rank_items(pipe.items, rank, world)
    items = len(mine)
    warm = min(warm_req, items - 1)
    it = pipe.run(rank=rank, world=world)
    for _ in range(warm):
        next(it)
    ctx.sync()
    if ctx.coll:
        dist.barrier()
    t0 = time.perf_counter()
    done, entropy, seen = 0, 0.0, []
    for item in it:
        entropy += float(item.cwt.stats[0, 1])  # (touch the result: the item is complete)
        seen.append((item.first_channel, item.chunk))
        done += 1
    ctx.sync()
    if ctx.coll:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    rank_dt = [dt_local]
    rank_items_done = [done]
    dt = dt_local
    if ctx.coll:
        t = torch.tensor([dt, float(done)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_dt = [float(v[0].item()) for v in every]
        rank_items_done = [int(v[1].item()) for v in every]
        dt = max(rank_dt)
    if stub and args.stub_dump:
        torch.save({"rank": rank, "items": mine, "timed": seen, "warm": warm}, f"{args.stub_dump}.rank{rank}")
    line = None
    if rank == 0:
        points_item = 2 * n_ch * n_b * n
        value = points_item * sum(rank_items_done) / dt / 1e6
        h2d_ms = None
        if not stub:  # plain host -> device copy rate of one item (pinned), for the PCIe share of a step
            x = torch.empty((n_ch, n), dtype=tdtype).pin_memory()
            d = torch.empty((n_ch, n), dtype=tdtype, device=dev)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(5):
                d.copy_(x, non_blocking=True)
            torch.cuda.synchronize()
            h2d_ms = round((time.perf_counter() - tc) / 5 * 1e3, 3)
            del x, d
        full_items = 128 * 131 / n_ch  # items one GPU of eight owns in the 24 h x 1024-channel job at this block size
        line = {
            "metric": "TFR Mpoints/sec (CWT+STX+entropy)", "value": round(value, 1), "unit": "Mpoints/s", "n_gpus": world,
            "steps": done, "warmup": warm, "ms_per_step": round(dt / max(done, 1) * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if not stub else "stub (no transform ran)",
            "config": {
                "workload": f"BASELINE configs[4] (bounded sample): float64 records streamed from the host, chunks of 2^{args.log2n} "
                            f"samples with a hop of 2^{args.log2n - 1} @ {fs:g} Hz, order N={order:g}, CWT+STX+entropy ({n_b} bands), "
                            f"reduced products only; {n_ch} records x {items} chunks per GPU here, 128 x 131 in the 24 h job",
                "channels_per_gpu": n_ch, "n": n, "bands": n_b, "points_per_step": points_item * world,
                "world_size": dist.get_world_size() if ctx.coll else 1, "backend": ctx.backend,
                "rank_seconds": [round(v, 6) for v in rank_dt], "rank_items": rank_items_done,
                "h2d_ms_per_item": h2d_ms,
                "projected_seconds_24h_1024ch_8gpu": round(full_items * dt / max(done, 1), 1),
                "projection_note": f"arithmetic: {full_items:g} items per GPU x the measured seconds per item over {done} timed items",
            },
            "step_roofline": {
                "required_bytes_per_step": int(2 * (n_ch * n * real_bytes + n_ch * (n_b + n) * real_bytes)),
                "note": "no panel is stored in streaming mode: the required HBM bytes are the records and the marginals only "
                        "(see the f64 leg / --dtype f64 for the stage breakdown of the float64 engines)",
            },
        }
        # No panel byte is stored here, so the HBM roofline does not bound this leg: its kernels are priced against the
        # double-precision vector issue rate.  Instructions per item = SQ_INSTS_VALU summed over the kernels of one item
        # (rocprofv3 --pmc pass of this command, profiles/), achieved = that count over the measured time per item.
        tdata = load_traffic()
        insts = tdata.get(f"valu_wave_insts_per_item:{args.dtype}:n{args.log2n}:o{order:g}:c{n_ch}:stream")
        item_ms = dt / max(done, 1) * 1e3
        ach = insts / (item_ms * 1e6) if insts else None  # wave instructions per nanosecond = G / s
        line["roofline"] = {
            "bound": "fp64-valu" if args.dtype == "f64" else "fp32-valu", "kernel": "all kernels of one streamed item",
            "achieved": round(ach, 2) if ach else None, "peak": FP64_VALU_PEAK_GINST,  # (float32 too: tools/micro/pk_rate.hip measures 4 cycles per wave64 v_fma_f32)
            "unit": "G wave-instructions/s", "frac": round(ach / FP64_VALU_PEAK_GINST, 4) if ach else None,
            "traffic": tdata.get(f"item:{args.dtype}:n{args.log2n}:o{order:g}:c{n_ch}:stream"),
            "valu_wave_instructions_per_item": insts,
            "valu_instructions_per_output": round(insts * 64 / points_item, 2) if insts else None,
            "source": (tdata.get("source_f64", "profiles/") + " (SQ_INSTS_VALU / TCC passes of this command, not collected in this run)") if insts else None,
            "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per double-precision wave64 instruction; every vector instruction is "
                    "counted as if it were double precision (an upper bound of the issue time the kernels need)",
        }
        if cpu:
            line["cpu_baseline"] = cpu
    plan.close()
    del pipe, plan, rows, host
    if not stub:
        import gc

        gc.collect()
        torch.cuda.empty_cache()
    return line


class StubPlan:
    """Stand-in for TfrPlan in --stub runs: no kernel, deterministic reduced products on the CPU."""

    def __init__(self, n, n_b, rank, dtype=None, step_ms=0.0):
        import torch

        self.n, self.n_b, self.rank, self.calls = n, n_b, rank, 0
        self.rdtype, self.device = dtype or torch.float32, torch.device("cpu")
        self.step_ms = step_ms

    def cwt_stx(self, sig, coef=True, bits=False, reductions=False, power_scale=1.0, eps=0.0, out=None, reduced_out=None):
        import torch
        from quantum_inferno_amd import dist as qdist
        from quantum_inferno_amd.engine import TfrResult

        n_ch = sig.shape[0]
        if self.step_ms > 0:
            time.sleep(self.step_ms * 1e-3)
        if out is None:
            out = []
            for k in range(2):
                r = TfrResult(frequency_hz=np.arange(self.n_b))
                r.reduced = reduced_out[k] if reduced_out and reduced_out[k] is not None else torch.empty(
                    qdist.reduced_slots(n_ch, self.n_b, self.n, sig.dtype), dtype=torch.float64)
                o1 = r.reduced.numel() - n_ch * (self.n_b + 4)
                o2 = o1 + n_ch * self.n_b
                r.power_time = r.reduced[:o1].view(sig.dtype)[: n_ch * self.n].view(n_ch, self.n)
                r.power_band = r.reduced[o1:o2].view(n_ch, self.n_b)
                r.stats = r.reduced[o2:].view(n_ch, 4)
                out.append(r)
            out = tuple(out)
        for k, r in enumerate(out):  # values that name the rank, the transform and the call
            r.power_band.fill_(1000.0 * self.rank + 100.0 * k + self.calls)
            r.power_time.fill_(float(self.rank + k))
            r.stats.fill_(float(self.calls))
        self.calls += 1
        return out

    def profile(self, *a, **k):
        pass

    def profile_read(self):
        return {"zoom": (1.0, 3), "block": (0.5, 3)}

    def stage_bands(self, stage):
        return [self.n_b // 2, 0, self.n_b // 2]

    def close(self):
        pass


class Ctx:
    """What every leg of one bench run shares: the rank layout, the device and the process group."""

    def __init__(self, world, rank, local, dev, stub, backend, coll=None):
        self.world, self.rank, self.local, self.dev, self.stub, self.backend = world, rank, local, dev, stub, backend
        # coll: a process group exists and every step ends in the gather (N > 1 ranks, or --force-collective on one)
        self.coll = (world > 1) if coll is None else coll

    def sync(self):
        if not self.stub:
            import torch

            torch.cuda.synchronize()


def leg_args(args, config, **over):
    """The arguments of one leg: BASELINE config `config` with the command line's overrides, then `over`."""
    a = argparse.Namespace(**vars(args))
    a.config = config
    for k, v in CONFIGS[config].items():
        if getattr(a, k, None) is None:
            setattr(a, k, v)
    for k, v in over.items():
        setattr(a, k, v)
    a.fs = 1000.0 if a.fs is None else a.fs
    a.dtype = a.dtype or "f32"
    a.stream = a.stream or 0
    small = a.channels * a.order <= 12
    if a.steps is None:
        a.steps = 200 if small else 20
    if a.warmup is None:
        a.warmup = 20 if small else 3
    return a


def fit_workspace(a, ctx, n, n_b, tdtype, real_bytes, depth):
    """Plan scratch for this leg, checked against the free HBM of this rank BEFORE anything is allocated: the two complex
    panels, the message buffers, rank 0's receive buffers (`depth` x world x message) and the STFT outputs are fixed
    costs; the scratch takes what is asked for (<= 48 GiB) or what is left -- fewer records per tile, never a failed
    allocation in the middle of the first multi-GPU run.  Returns (workspace bytes, budget dict)."""
    import torch

    import quantum_inferno_amd as qi
    from quantum_inferno_amd import dist as qdist

    n_ch = a.channels
    want = int(a.workspace_gib * 2 ** 30) if a.workspace_gib > 0 else qi.TfrPlan.workspace_for(n, n_b, tdtype, n_ch, cap_bytes=48 << 30)
    if ctx.stub:
        return want, {}
    free, total = torch.cuda.mem_get_info(ctx.dev)
    msg = 2 * qdist.reduced_slots(n_ch, n_b, n, tdtype) * 8
    panels = 2 * n_ch * n_b * n * 2 * real_bytes
    recv = depth * ctx.world * msg if (ctx.coll and ctx.rank == 0) else 0
    stft = 0
    if a.stft:
        seg = 2048 if a.order >= 12 else 512  # (an upper bound is enough here)
        stft = int(n_ch * (seg // 2 + 1) * (n // (seg // 2) + 2) * 3 * real_bytes)
    fixed = panels + depth * msg + recv + stft + n_ch * n * real_bytes + (6 << 30)  # + bank tables, allocator slack
    room = free - fixed
    per_rec = qi.TfrPlan.workspace_for(n, n_b, tdtype, 1, cap_bytes=0)
    budget = {"free_hbm_bytes": int(free), "panels_bytes": int(panels), "messages_bytes": int(depth * msg), "receive_bytes": int(recv),
              "scratch_bytes": int(want)}
    if room < per_rec:
        raise RuntimeError(f"rank {ctx.rank}: {n_ch} records x {n_b} bands x 2^{a.log2n} need {fixed / 2**30:.1f} GiB + scratch, "
                           f"{free / 2**30:.1f} GiB are free: lower --channels")
    if want > room:
        want = int(room)
        budget["scratch_bytes"] = want
        budget["note"] = "scratch cut to the free HBM: the records pass in more tiles"
    return want, budget


def run_leg(a, ctx, cpu, extras=True):
    """One timed leg.  Returns the JSON record (rank 0) or None."""
    import torch
    import torch.distributed as dist

    import quantum_inferno_amd as qi
    from quantum_inferno_amd import _lib, dist as qdist, styx_fft, synth

    world, rank, dev, stub = ctx.world, ctx.rank, ctx.dev, ctx.stub
    device_sync = ctx.sync
    n, fs, order = 1 << a.log2n, a.fs, a.order
    tdtype = torch.float32 if a.dtype == "f32" else torch.float64
    real_bytes = 4 if a.dtype == "f32" else 8
    n_ch = a.channels
    total_ch = n_ch * world
    first, _ = qdist.shard(total_ch, rank, world)
    bands = qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
    n_b = len(bands)
    engine_code = {"auto": _lib.QI_ENGINE_AUTO, "hipfft": _lib.QI_ENGINE_HIPFFT, "native": _lib.QI_ENGINE_NATIVE}[a.engine]
    depth = 2 if ctx.coll else 1
    ws, budget = fit_workspace(a, ctx, n, n_b, tdtype, real_bytes, depth)
    if stub:
        plan = StubPlan(n, n_b, rank, step_ms=a.stub_ms)
        sig = torch.zeros((n_ch, n), dtype=tdtype)
        stft = None
    else:
        plan = qi.TfrPlan(n, tdtype, dev, ws, engine_code)
        plan.set_styx_bank(order, fs)
        plan.set_stx_bands(order, fs)
        sig = torch.from_numpy(synth.channels(n, fs, first, n_ch, total_ch, np.float32 if tdtype == torch.float32 else np.float64)).to(dev)
        stft = styx_fft.StftPlan(n, n_ch, fs, order, tdtype, dev) if a.stft else None

    # the reduced products of both transforms live in one buffer: the message of the gather, no packing copy.  With more
    # than one rank the gather of step k overlaps the transforms of step k + 1 (two sets of outputs, used in turn); the
    # complex panels themselves are written in place every step (one set: 2 x 89.7 GB at configs[2] / [3]).
    slots = qdist.reduced_slots(n_ch, n_b, n, tdtype)
    messages = [torch.empty(2 * slots, dtype=torch.float64, device=dev) for _ in range(depth)]
    outs = [plan.cwt_stx(sig, coef=True, reductions=True, reduced_out=(messages[0][:slots], messages[0][slots:]))]
    for m in messages[1:]:  # further message buffers share the panels of the first set
        oc, os_ = plan.cwt_stx(sig, coef=False, reductions=True, reduced_out=(m[:slots], m[slots:]))
        oc.coef, os_.coef = outs[0][0].coef, outs[0][1].coef  # (None in a --stub run)
        outs.append((oc, os_))
    small = n_ch * order <= 12  # a step of a quarter of a millisecond: every recorded event shows
    pipe = qdist.GatherPipeline(depth=depth, dst=0, timing=ctx.coll and not small, force_collective=ctx.coll)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if not stub else None
    stft_ms = []

    def step(time_stft=False):
        if stft is not None:
            if time_stft:
                ev[0].record()
            stft.run(sig)
            if time_stft:
                ev[1].record()
        if not ctx.coll:
            plan.cwt_stx(sig, out=outs[0])  # qi_cwt_stx: both transforms of the same records in one call
            msg = qdist.pack_reduced(list(outs[0]))
        else:
            i = pipe.acquire()
            plan.cwt_stx(sig, out=outs[i])
            msg = pipe.submit(i, qdist.pack_reduced(list(outs[i])))
        if time_stft and stft is not None:
            device_sync()
            stft_ms.append(ev[0].elapsed_time(ev[1]))
        return msg

    last_gathered = []

    def fence():
        last_gathered[:] = pipe.drain()
        device_sync()
        if ctx.coll:
            dist.barrier()
        device_sync()

    t_warm = time.perf_counter()
    for _ in range(a.warmup):
        step()
    # a step of configs[1] is a quarter of a millisecond: a few warmup steps end long before the clocks have left idle
    settle_steps = 0
    while True:
        device_sync()
        done = (time.perf_counter() - t_warm) * 1e3 >= a.settle_ms
        if ctx.coll:  # every rank runs the same number of steps (each step ends in a collective)
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
        step(time_stft=True)
    device_sync()
    stage_all = plan.profile_read()

    # (of the stages that PRODUCE panel rows: block, zoom, pass2, inverse -- pass 1 of the two-pass engine, the forward
    # transform and the coarse stage write no coefficient)
    def produces(name):
        try:
            sb = plan.stage_bands(name)
        except ValueError:
            return False
        return sb[0] + sb[2] > 0

    producing = {k: v for k, v in stage_all.items() if v[1] and produces(k)}
    dominant = max((producing or stage_all).items(), key=lambda kv: kv[1][0])[0]
    # timed region: HIP events around the dominant stage's launches only, on every 7th transform call (odd, so that the
    # CWT and the Stockwell calls of a step are sampled alike) -- every event is a bubble in the stream.  A step of many
    # records is tens of milliseconds: there every producing stage is timed on every call.
    timed_stages = [dominant] if small else (list(producing) or [dominant])
    plan.profile(True, stages=timed_stages, period=7 if small else 1)
    pipe.reset_timing()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    stage = plan.profile_read()
    plan.profile(False)
    rank_dt = [dt_local]
    wait_ms = [pipe.wait_ms() / max(a.steps, 1)]
    n_ranks = 1
    # every rank's stage times (HIP events, the 3 untimed steps with every stage timed): a slow rank 0 -- RCCL's copy kernels
    # share its CUs with the transforms -- then shows as longer kernels THERE, apart from the time spent waiting for a gather
    stage_names = sorted(k for k, v in stage_all.items() if v[1])
    rank_stage_ms = {k: [round(stage_all[k][0] / 3, 4)] for k in stage_names}
    if ctx.coll:
        n_ranks = dist.get_world_size()
        from quantum_inferno_amd._lib import STAGES as all_stages

        t = torch.tensor([dt, wait_ms[0]] + [stage_all.get(k, (0.0, 0))[0] / 3 for k in all_stages], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_dt = [float(v[0].item()) for v in every]
        wait_ms = [float(v[1].item()) for v in every]
        rank_stage_ms = {k: [round(float(v[2 + i].item()), 4) for v in every] for i, k in enumerate(all_stages)
                         if any(float(v[2 + i].item()) > 0 for v in every)}
        dt = max(rank_dt)

    line = None
    if rank == 0:
        points_step = 2 * total_ch * n_b * n
        value = points_step * a.steps / dt / 1e6
        tdata = load_traffic()

        def stage_roofline(name):
            """Algorithmic bytes of a stage -- the complex coefficients of the bands it produces, written once (SURVEY
            s8d: C*B*n*s_c), plus the per-time / per-band marginals it leaves behind -- per launch over the mean launch
            duration: from the timed region when the stage was timed there, else from the three untimed steps."""
            timed = name in timed_stages and stage[name][1] > 0
            ms, launches = stage[name] if timed else stage_all[name]
            per_launch_ms = ms / max(launches, 1)
            launches_per_step = max(stage_all[name][1] / 3, 1)
            sb = plan.stage_bands(name)
            nb_stage = sb[0] + sb[2]  # styx CWT + Stockwell panels of one step
            alg_step = n_ch * nb_stage * n * 2 * real_bytes + 2 * n_ch * n * real_bytes + n_ch * nb_stage * real_bytes
            alg_launch = alg_step / launches_per_step
            achieved = alg_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
            traffic = tdata.get(f"{name}:{a.dtype}:n{a.log2n}:o{order:g}:c{n_ch}")
            # the other ceiling of these kernels: vector wave instructions per launch (SQ_INSTS_VALU of the builder's counter
            # pass, like `traffic`) x 4 cycles per wave64 instruction over the SIMD-cycles of the measured launch
            valu = tdata.get(f"valu:{name}:{a.dtype}:n{a.log2n}:o{order:g}:c{n_ch}")
            valu_issue = None
            if valu is not None and per_launch_ms > 0:
                valu_issue = {"wave_insts_per_launch": int(valu), "cycles_per_inst": 4, "simds": 1024, "clock_ghz": 2.4,
                              "frac": round(valu * 4 / (1024 * per_launch_ms * 1e-3 * 2.4e9), 4),
                              "source": tdata.get("source_valu")}
            return {
                "bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (tdata.get("source", "profiles/traffic.json") + " (rocprofv3 --pmc passes of this command kept "
                                   "under profiles/, not collected in this run)") if traffic is not None else None,
                "launch_ms": round(per_launch_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": int(alg_launch), "bands_per_step": nb_stage,
                "timed_in": "timed region (HIP events)" if timed else "3 untimed steps after the warmup (HIP events)",
                "valu_issue": valu_issue,
            }

        dev_ms = sum(v[0] for v in stage_all.values()) / 3
        req = 2 * required_bytes(n_ch, n_b, n, real_bytes)
        stft_info = None
        if stft is not None:
            stft_alg = n_ch * n * real_bytes + stft.points * 2 * real_bytes + stft.points * real_bytes  # records in, Z and bits out
            req += stft_alg
            s_ms = float(np.median(stft_ms)) if stft_ms else 0.0
            stft_info = {
                "segment": stft.seg, "shape_per_channel": [stft.n_f, stft.n_seg], "points_per_step": stft.points * world,
                "ms_per_step": round(s_ms, 4), "algorithmic_bytes_per_step": int(stft_alg),
                "achieved_gbs": round(stft_alg / (s_ms * 1e-3) / 1e9, 1) if s_ms > 0 else None,
                "frac": round(stft_alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if s_ms > 0 else None,
            }
            dev_ms += s_ms
        wall_ms = dt / a.steps * 1e3
        base = CONFIGS[a.config]
        cfg_name = f"BASELINE configs[{a.config}]"
        if (n_ch, order, bool(a.stft), a.dtype) != (base["channels"], base["order"], bool(base["stft"]), base.get("dtype", "f32")):
            cfg_name += " (modified)"
        if world > 1 and a.config == 2 and "(modified)" not in cfg_name:
            cfg_name = f"BASELINE configs[3] ({total_ch} channels sharded over {world} GPUs = configs[2] per GPU)"
        line = {
            "metric": "TFR Mpoints/sec (CWT+STX+entropy)",
            "value": round(value, 1),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(wall_ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}: {n_ch} channel(s) per GPU x 2^{a.log2n} samples @ {fs:g} Hz, "
                            f"order N={order:g}, {'STFT+' if stft is not None else ''}CWT+STX+entropy, {n_b} bands",
                "channels_per_gpu": n_ch,
                "n": n,
                "bands": n_b,
                "points_per_step": points_step,
                "engine": a.engine,
                "world_size": n_ranks,
                "backend": ctx.backend,
                "rank_seconds": [round(v, 6) for v in rank_dt],
                "gather_wait_ms_per_step": [round(v, 4) for v in wait_ms] if pipe.timing else None,
                "gather_message_bytes_per_rank": int(2 * slots * 8) if ctx.coll else 0,
                "rank_stage_ms_per_step": rank_stage_ms if ctx.coll else None,
            },
            "roofline": stage_roofline(dominant),
            "step_roofline": {
                "required_bytes_per_step": int(req),
                "wall_ms_per_step": round(wall_ms, 4),
                "achieved_gbs": round(req / (wall_ms * 1e-3) / 1e9, 1),
                "frac": round(req / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "device_ms_per_step": round(dev_ms, 4),
                "stage_ms_per_step": {k: round(v[0] / 3, 4) for k, v in stage_all.items() if v[1]},
                "survey_bytes_per_step": int(survey_bytes(n_ch, n_b, n, 2 * n, real_bytes)),
                "note": "required bytes = records in + complex panels out + marginals out (+ the STFT's in / out) over the "
                        "timed wall step; stage breakdown from 3 untimed steps with every stage under HIP events; "
                        "survey_bytes adds SURVEY s8(d)'s atom-bank read, which this engine never performs",
            },
        }
        others = [k for k in producing if k != dominant]
        if others:
            line["stage_rooflines"] = {k: stage_roofline(k) for k in others}
        if budget:
            line["config"]["hbm_budget"] = budget
        if stft_info:
            line["stft"] = stft_info
        if extras and world == 1 and not stub and a.wrappers and n_ch == 1:
            # the drop-in call as the tutorials make it (s04_tone_tfr.py:84-99): NumPy in, NumPy out, one transform per
            # call -- host <-> device copies and the widening to the reference's complex128 included (never `value`)
            from quantum_inferno_amd import engine as qengine, styx_cwt, styx_stx

            # A caller that consumes a result and drops it before the next call (a loop over records): the page-locked
            # block of the result goes back to PyTorch's host allocator and is handed out again, so the steady state is one
            # device-to-host copy at PCIe speed per transform.  The FIRST call of a process also page-locks the blocks (and
            # builds the plan): reported apart.  (Round 3 timed the second of two calls while the first one's results were
            # still alive -- a fresh 0.8 GB page-locked allocation inside the timed call.)
            x_host = sig[0].cpu().numpy()
            out_w = {}
            for mode in ("reference", "native"):
                qengine.NUMPY_RESULT_DTYPE = mode
                times = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    tw0 = time.perf_counter()
                    c_np = styx_cwt.cwt_complex_any_scale_pow2(order, x_host, fs)[2]
                    s_np = styx_stx.stx_complex_any_scale_pow2(order, x_host, fs)[2]
                    times.append(time.perf_counter() - tw0)
                    dtype_name, host_bytes = str(c_np.dtype), int(c_np.nbytes + s_np.nbytes)
                    del c_np, s_np
                tw = float(np.median(times[1:]))
                out_w[mode] = {"ms": round(tw * 1e3, 2), "first_call_ms": round(times[0] * 1e3, 2),
                               "mpoints_per_s": round(2 * n_b * n / tw / 1e6, 1), "result_dtype": dtype_name,
                               "host_bytes": host_bytes, "d2h_gbs": round(host_bytes / tw / 1e9, 1),
                               "note": "median of 4 calls after the first, every result dropped before the next call"}
            qengine.NUMPY_RESULT_DTYPE = "reference"
            qengine.clear_plans()
            line["numpy_wrappers"] = out_w
        if extras and world == 1 and not stub and a.two_streams and n_ch <= 2 and stft is None and a.dtype == "f32":
            # Calls of one or two records: a fifth of the step are short launches (forward transform, coarse stage, tail)
            # that leave most of the chip idle.  A caller that has independent records to transform hides them by
            # alternating two plans on two streams (no cross-stream events inside a step) -- reported beside `value`,
            # never as `value`: the kernels' durations stretch while they share the chip, so the roofline above would
            # not describe them.
            plans2 = [plan, qi.TfrPlan(n, tdtype, dev, ws, engine_code)]
            plans2[1].set_styx_bank(order, fs)
            plans2[1].set_stx_bands(order, fs)
            streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            outs2 = [outs[0], plans2[1].cwt_stx(sig, coef=True, reductions=True)]
            torch.cuda.synchronize()

            def run2(k):
                for i in range(k):
                    j = i & 1
                    with torch.cuda.stream(streams2[j]):
                        plans2[j].cwt_stx(sig, out=outs2[j])
            run2(200)
            torch.cuda.synchronize()
            k2 = 1000
            t2 = time.perf_counter()
            run2(k2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            line["two_streams"] = {"steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 4),
                                   "value": round(points_step * k2 / dt2 / 1e6, 1),
                                   "note": "the same steps issued alternately on two streams (two plans, two sets of buffers): the "
                                           "short launches of one step run under the long launches of the other"}
            plans2[1].close()
            del plans2, outs2
        if cpu:
            line["cpu_baseline"] = cpu
        if stub:
            line["data"] = "stub (CPU rehearsal of the rank plumbing, no transform ran)"
            if a.stub_dump:
                gathered = [g.clone() if g is not None else None for g in last_gathered] if ctx.coll else [qdist.pack_reduced(list(outs[0])).unsqueeze(0)]
                torch.save({"gathered": gathered, "slots": slots, "n_ch": n_ch, "n_b": n_b, "n": n, "calls": plan.calls}, a.stub_dump)
    plan.close()
    # hand every buffer of this leg back before the next one sizes itself (configs[2] holds 2 x 89.7 GB of panels)
    del outs, messages, sig, stft, plan, pipe, last_gathered
    qdist.clear_gather_buffers()
    if not stub:
        import gc

        gc.collect()
        torch.cuda.empty_cache()
    return line


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    shaped = any(getattr(args, k) is not None for k in ("channels", "order", "stft", "dtype", "stream"))
    composite = args.config is None and not shaped
    # The legs of this run.  No --config and no shape flag:
    #   one rank   -> `value` = BASELINE configs[1] (the config the metric is quoted on at one GPU), and in the same process,
    #                 under their own keys, configs[2] (64 records x order 12, the full stack) and a float64 leg;
    #   N ranks    -> `value` = configs[3]: configs[2]'s 64 records per GPU, sharded, one gather of the reduced product per
    #                 step; the one-record-per-GPU latency case rides along as `configs1_per_gpu`.
    if not composite:
        legs = [("", leg_args(args, args.config or 1), args.cpu_seconds)]
    elif world == 1:
        legs = [("", leg_args(args, 1), args.cpu_seconds),
                ("configs2", leg_args(args, 2), min(args.cpu_seconds, 12.0)),
                ("f64", leg_args(args, 2, channels=4, dtype="f64", stft=0), min(args.cpu_seconds, 9.0)),
                ("configs4", leg_args(args, 4), min(args.cpu_seconds, 9.0))]
    else:
        legs = [("", leg_args(args, 2), 0.0), ("configs1_per_gpu", leg_args(args, 1), 0.0)]
    if args.legs:
        keep = set(args.legs.split(","))
        legs = [l for l in legs if (l[0] or "main") in keep]
    # CPU baselines first: the forked workers must exist before anything initialises the GPU
    cpus = []
    for key, a, budget in legs:
        if world == 1 and budget > 0 and not a.stub:
            a_cpu = argparse.Namespace(**vars(a))
            a_cpu.cpu_seconds = budget
            cpus.append(cpu_baseline(a_cpu, 1 << a.log2n, a.fs, a.order))
        else:
            cpus.append(None)

    import torch
    import torch.distributed as dist

    stub = bool(args.stub)
    backend = "none"
    coll = world > 1 or bool(args.force_collective)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (a forced one-rank group outside torch.distributed.run)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        backend = dist.get_backend() + (" (RCCL over xGMI)" if not stub else "")
    if stub:
        dev = torch.device("cpu")
        torch.set_num_threads(1)  # (as under torch.distributed.run: the stub's small fills otherwise pay for a thread pool)
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    ctx = Ctx(world, rank, local, dev, stub, backend, coll)

    line = None
    for (key, a, _), cpu in zip(legs, cpus):
        if a.stream:
            rec = stream_bench(a, ctx, cpu)
        else:
            rec = run_leg(a, ctx, cpu, extras=(key == ""))
        if rank == 0:
            if line is None:
                line = rec
                if key:
                    line["leg"] = key
            else:
                line[key] = rec
    if rank == 0:
        if composite:
            # `value` of an N-rank default run is BASELINE configs[3] (64 records per GPU), `value` of the one-rank default run is
            # configs[1] (one record): a scaling curve must start from `scaling_base`, not from the one-rank `value`
            line["scaling_note"] = ("weak scaling of BASELINE configs[3] (64 records per GPU, order 12, STFT+CWT+STX+entropy): with N > 1 "
                                    "ranks `value` is that workload; the one-GPU point of the curve is `scaling_base` of the N = 1 line "
                                    "(= its `configs2` record; the N = 1 `value` itself is configs[1], one record)")
            base = line.get("configs2") if world == 1 else None
            if base:
                line["scaling_base"] = {"workload": base["config"]["workload"], "value": base["value"], "unit": base["unit"],
                                        "ms_per_step": base["ms_per_step"], "n_gpus": base["n_gpus"],
                                        "channels_per_gpu": base["config"]["channels_per_gpu"],
                                        "points_per_step": base["config"]["points_per_step"]}
        print(json.dumps(line), flush=True)
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
