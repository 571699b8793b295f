#!/usr/bin/env python3
"""Diagnostic: run one transform with the stamped library (python quantum_inferno_amd/_build.py --stamps) and let it
print the mean cycles each block / pass-2 workgroup spent per phase (at plan destruction).
usage: QI_TUNE=1 QI_TFR_LIB=.../libqi_tfr_stamps.so QI_NATIVE_STAMPS=1 python tools/stamps.py [cwt|stx] [order] [channels] [f32|f64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_inferno_amd as qi
from quantum_inferno_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "stx"
order = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n_ch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n, fs = 1 << 20, 1000.0
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
plan = qi.TfrPlan(n, dt, "cuda:0", qi.TfrPlan.workspace_for(n, nb, dt, n_ch, cap_bytes=32 << 30))
sig = torch.from_numpy(synth.channels(n, fs, 0, n_ch, n_ch, np.float64 if dt == torch.float64 else np.float32)).cuda()
if which == "cwt":
    plan.set_styx_bank(order, fs)
    for _ in range(3): out = plan.cwt(sig, coef=True, reductions=True)
else:
    plan.set_stx_bands(order, fs)
    for _ in range(3): out = plan.stx(sig, coef=True, reductions=True)
torch.cuda.synchronize()
print(which, "done", flush=True)
plan.close()
