"""Single-record steps of configs[1] through engine.PlanRing (two plans on two streams, used in turn) against one plan on
one stream: do the short launches of one step hide under the long launches of the other?  (GPU box.)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantum_inferno_amd as qi
from quantum_inferno_amd import synth, scales_dyadic

n, fs, order = 1 << 20, 1000.0, 3
nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
x = torch.from_numpy(synth.log_chirp(n, fs, 0, 1, np.float32)).cuda().unsqueeze(0)
setup = lambda p: (p.set_styx_bank(order, fs), p.set_stx_bands(order, fs))
for depth, wait in ((1, False), (2, False), (2, True), (3, False), (1, False), (2, False)):
    ring = qi.PlanRing(n, torch.float32, setup=setup, depth=depth, wait_input=wait)
    for _ in range(200):
        ring.cwt_stx(x, coef=True, reductions=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        ring.cwt_stx(x, coef=True, reductions=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"depth {depth}, wait_input {wait}: {dt:.4f} ms per step, {2 * nb * n * 1000 / dt / 1e6:.0f} Mpoints/s")
    ring.close()
