#!/usr/bin/env python3
"""qi_cwt_stx of one record as a captured graph (QI_PLAN_GRAPH) against the eager launches: identical results, time per step.
usage (GPU box): python tools/graph_probe.py [channels]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import synth  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, fs, order = 1 << 20, 1000.0, 3
nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
sig = torch.from_numpy(synth.channels(n, fs, 0, C, C, np.float32)).cuda()
res = {}
for name, graph in (("eager", False), ("graph", True)):
    plan = qi.TfrPlan(n, torch.float32, "cuda:0", qi.TfrPlan.workspace_for(n, nb, torch.float32, C), graph=graph)
    plan.set_styx_bank(order, fs)
    plan.set_stx_bands(order, fs)
    out = plan.cwt_stx(sig, coef=True, reductions=True)
    for _ in range(300):
        plan.cwt_stx(sig, out=out)
    torch.cuda.synchronize()
    time.sleep(0.05)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            plan.cwt_stx(sig, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    print(f"{name}: {best * 1e3:.4f} ms per step = {2 * C * nb * n / best / 1e6:.0f} Mpoints/s")
    res[name] = [t.clone() for r in out for t in (r.coef, r.reduced)]
    plan.close()
same = all(torch.equal(a, b) for a, b in zip(res["eager"], res["graph"]))
print("graph results bit-equal to the eager launches:", same)
sys.exit(0 if same else 1)
