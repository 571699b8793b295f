#!/usr/bin/env python3
"""Records shorter than 2^15 samples on the zoom / block engines (QI_NATIVE_MIN_LOG2N) against the hipFFT engine: which bands
each engine takes, every row's error, time per call of both.  usage (GPU box): python tools/small_n_probe.py [min_log2n]"""
import os
import sys
import time

os.environ["QI_TUNE"] = "1"
os.environ["QI_NATIVE_MIN_LOG2N"] = sys.argv[1] if len(sys.argv) > 1 else "12"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import _lib, synth  # noqa: E402


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for dt, tol in ((torch.float32, 2e-5), (torch.float64, 1e-10)):
    for log2n in (12, 13, 14):
        for order in (3, 12):
            n, fs = 1 << log2n, 1000.0
            nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
            npd = np.float64 if dt == torch.float64 else np.float32
            x = torch.from_numpy(synth.channels(n, fs, 0, 2, 2, npd) + 0.2 * np.random.default_rng(log2n).standard_normal((2, n)).astype(npd)).cuda()
            ws = qi.TfrPlan.workspace_for(n, nb, dt, 2)
            nat = qi.TfrPlan(n, dt, "cuda:0", ws, _lib.QI_ENGINE_AUTO)
            ref = qi.TfrPlan(n, dt, "cuda:0", ws, _lib.QI_ENGINE_HIPFFT)
            line = f"{'f64' if dt == torch.float64 else 'f32'} 2^{log2n} order {order:2d} ({nb:3d} bands):"
            for plan in (nat, ref):
                plan.set_styx_bank(order, fs)
                plan.set_stx_bands(order, fs)
            for which, name in ((0, "cwt"), (2, "stx")):
                a = getattr(nat, name)(x, coef=True, reductions=True)
                b = getattr(ref, name)(x, coef=True, reductions=True)
                rows = (a.coef - b.coef).abs().amax(dim=2) / b.coef.abs().amax(dim=2)
                red = float(((a.power_band - b.power_band).abs() / b.power_band).max())
                tn, tr = timed(lambda: getattr(nat, name)(x, out=a)), timed(lambda: getattr(ref, name)(x, out=b))
                line += (f"  {name}: zoom {nat.stage_bands('zoom')[which]} block {nat.stage_bands('block')[which]} two-pass {nat.stage_bands('pass2')[which]}"
                         f" hipfft {nat.stage_bands('inverse')[which]} | worst row {float(rows.max()):.1e} band power {red:.1e}"
                         f" {'OK' if float(rows.max()) <= tol else 'FAIL'} | {tn:.0f} us vs {tr:.0f} us")
            print(line, flush=True)
            nat.close()
            ref.close()
