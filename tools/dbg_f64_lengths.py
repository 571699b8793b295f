#!/usr/bin/env python3
"""Which engine takes the bands of a float64 table at a given length (QI_TUNE=1 QI_NATIVE_VERBOSE=1 prints the plan's choices)."""
import os, sys
os.environ["QI_TUNE"] = "1"
os.environ["QI_NATIVE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_inferno_amd as qi
from quantum_inferno_amd import engine
for log2n, name, order in ((16, "stx", 3), (16, "cwt", 12), (15, "cwt", 6)):
    n = 1 << log2n
    nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(1000.0, n, order))
    p = engine.TfrPlan(n, np.float64, None, engine.TfrPlan.workspace_for(n, nb, np.float64, 1))
    print(f"== 2^{log2n} {name} order {order}: {nb} bands", file=sys.stderr, flush=True)
    (p.set_styx_bank if name == "cwt" else p.set_stx_bands)(order, 1000.0)
    w = 0 if name == "cwt" else 2
    print("   stage bands zoom/block/pass2:", p.stage_bands("zoom")[w], p.stage_bands("block")[w], p.stage_bands("pass2")[w], file=sys.stderr, flush=True)
    p.close()
