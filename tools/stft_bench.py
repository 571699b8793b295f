#!/usr/bin/env python3
"""Time the fused STFT alone (configs[2] shape by default): tools/stft_bench.py [channels] [order] [log2n] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantum_inferno_amd import styx_fft, synth  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
order = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 20)
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
x = torch.from_numpy(synth.channels(n, 1000.0, 0, ch, ch, np.float64 if dt == torch.float64 else np.float32)).cuda()
plan = styx_fft.StftPlan(n, ch, 1000.0, order, dt)
for _ in range(5):
    plan.run(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 30
e0.record()
for _ in range(reps):
    plan.run(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
rb = 4 if dt == torch.float32 else 8
alg = ch * n * rb + plan.points * 3 * rb
print(f"stft {ch} x 2^{int(np.log2(n))} order {order:g} seg {plan.seg}: {ms:.4f} ms, {alg / ms / 1e6:.0f} GB/s, frac {alg / ms / 1e6 / 8000:.3f}")
