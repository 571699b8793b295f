#!/usr/bin/env python3
"""Randomised parity stress (GPU box): the native engines against the ORACLE (oracle/tfr_oracle.py, float64 NumPy) at random
power-of-two lengths 2^14 .. 2^18, band orders, sample rates, batch sizes and both precisions -- five random bands of each
panel of a random record of the batch, their powers, and the entropy / total of the whole panel.  Fixed seed; prints one line
per case and the worst errors.  tools/stress_oracle.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tfr_oracle as orc  # noqa: E402
from quantum_inferno_amd import engine, scales_dyadic  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
worst = {np.float32: 0.0, np.float64: 0.0}
bad = 0
for case in range(cases):
    log2n = int(rng.integers(14, 19))  # (round 5: from 2^14, where float32 records go native)
    order = float(rng.choice([1, 2, 3, 4, 6, 8, 12]))
    fs = float(rng.choice([200.0, 800.0, 1000.0, 8000.0]))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    C = int(rng.integers(1, 6))
    n = 1 << log2n
    nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
    x = np.stack([orc.synth_chirp(n, fs, c, C, np.float64) for c in range(C)]) + 0.2 * rng.standard_normal((C, n))
    x = x.astype(dtype)
    plan = engine.TfrPlan(n, dtype, None, engine.TfrPlan.workspace_for(n, nb, dtype, C))
    plan.set_styx_bank(order, fs)
    plan.set_stx_bands(order, fs)
    xt = torch.from_numpy(x).cuda()
    rc, rs = plan.cwt_stx(xt, coef=True, reductions=True)
    c = int(rng.integers(0, C))
    pick = sorted(set(int(v) for v in rng.integers(0, nb, 5)))
    line = f"case {case:2d}: 2^{log2n} order {order:g} fs {fs:g} {np.dtype(dtype).name} C {C} bands {nb} engines zoom/block/pass2 " \
           f"{plan.stage_bands('zoom')} {plan.stage_bands('block')} {plan.stage_bands('pass2')}"
    tol = 2e-5 if dtype == np.float32 else 5e-9
    for name, res, fn in (("cwt", rc, orc.cwt_fft), ("stx", rs, orc.stx_fft)):
        _, _, want = fn(order, x[c].astype(np.float64), fs, bands=pick)
        got = res.coef[c][torch.tensor(pick, device="cuda")].cpu().numpy()
        err = max(float(np.max(np.abs(got[i] - want[i])) / np.max(np.abs(want[i]))) for i in range(len(pick)))
        pb = float(np.max(np.abs(res.power_band[c][torch.tensor(pick, device="cuda")].cpu().numpy() / (np.abs(want) ** 2).sum(axis=1) - 1.0)))
        ok = err <= tol and pb <= (2e-4 if dtype == np.float32 else 1e-9)
        bad += 0 if ok else 1
        worst[dtype] = max(worst[dtype], err)
        line += f" | {name} row {err:.1e} band {pb:.1e}{'' if ok else ' FAIL'}"
    print(line, flush=True)
    plan.close()
    del xt, rc, rs
    torch.cuda.empty_cache()
print(f"worst row error relative to the row's own maximum: float32 {worst[np.float32]:.2e}, float64 {worst[np.float64]:.2e}; failures: {bad}")
sys.exit(1 if bad else 0)
