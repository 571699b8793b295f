#!/usr/bin/env python3
"""Per-band error of the float64 native engines against the hipFFT engine (the reference's algorithm on the GPU):
tools/dbg_f64_rows.py order [log2n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import _lib, engine, synth  # noqa: E402

order = float(sys.argv[1])
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
fs = 1000.0
x = torch.from_numpy(synth.log_chirp(n, fs, 0, 1, np.float64)).cuda().unsqueeze(0)
f = qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
ws = engine.TfrPlan.workspace_for(n, len(f), np.float64, 1)
nat = engine.TfrPlan(n, np.float64, None, ws, _lib.QI_ENGINE_NATIVE)
ref = engine.TfrPlan(n, np.float64, None, ws, _lib.QI_ENGINE_HIPFFT)
for pl in (nat, ref):
    pl.set_styx_bank(order, fs)
    pl.set_stx_bands(order, fs)
for name in ("cwt", "stx"):
    a = getattr(nat, name)(x, coef=True).coef[0]
    b = getattr(ref, name)(x, coef=True).coef[0]
    scale = float(b.abs().max())
    err = (a - b).abs().amax(dim=1).cpu().numpy()
    own = b.abs().amax(dim=1).cpu().numpy()
    worst = np.argsort(-err)[:12]
    print(name, "panel max", scale, "worst bands (band, err/panel max, err/own max, f Hz):")
    for j in worst:
        print(f"   {j:4d} {err[j] / scale:.2e} {err[j] / own[j]:.2e} {f[j]:.3f}")
