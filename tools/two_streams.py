"""Experiment: single-record steps of configs[1] issued alternately on two streams (two plans, two sets of buffers)
against one stream -- do the short launches of one step hide under the long launches of the other?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_inferno_amd import engine, synth, scales_dyadic

n, fs, order = 1 << 20, 1000.0, 3
nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
x = torch.from_numpy(synth.log_chirp(n, fs, 0, 1, np.float32)).cuda().unsqueeze(0)
def make():
    p = engine.TfrPlan(n, torch.float32, None, engine.TfrPlan.workspace_for(n, nb, torch.float32, 1))
    p.set_styx_bank(order, fs); p.set_stx_bands(order, fs)
    return p
plans = [make(), make()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]
def run(k, ns):
    for i in range(k):
        j = i % ns
        with torch.cuda.stream(streams[j]):
            outs[j] = plans[j].cwt_stx(x, coef=True, reductions=True, out=outs[j])
for ns in (1, 2, 1, 2):
    run(200, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(1000, ns); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {dt / 1000 * 1e3:.4f} ms per step, {2 * nb * n * 1000 / dt / 1e6:.0f} Mpoints/s")
