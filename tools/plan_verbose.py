#!/usr/bin/env python3
"""Print the band classification of a plan (QI_TUNE=1 QI_NATIVE_VERBOSE=1 is set here): tools/plan_verbose.py f64|f32 order [log2n]"""
import os
import sys

os.environ["QI_TUNE"] = "1"
os.environ["QI_NATIVE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantum_inferno_amd as qi  # noqa: E402

dt = torch.float64 if sys.argv[1] == "f64" else torch.float32
order = float(sys.argv[2])
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 20)
f = qi.scales_dyadic.log_frequency_hz_from_fft_points(1000.0, n, order)
plan = qi.TfrPlan(n, dt, "cuda:0", qi.TfrPlan.workspace_for(n, len(f), dt, 1))
plan.set_styx_bank(order, 1000.0)
plan.set_stx_bands(order, 1000.0)
for st in ("pass1", "pass2", "block", "zoom"):
    print(st, plan.stage_bands(st))
