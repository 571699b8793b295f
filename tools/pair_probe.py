#!/usr/bin/env python3
"""qi_cwt_stx of one record (configs[1]) with the joint block launch on a side stream (QI_NATIVE_PAIR 1 ... 5; round 5 also built 6 / 7 = a CU-masked stream,
profiles/r05_cumask_timeline.txt, and removed it again: QI_NATIVE_PAIR=6 / 7,
QI_NATIVE_PAIR_CUS, QI_NATIVE_PAIR_MAIN) against the serial launch order: time per step and bit-equality of every output.
usage (GPU box): QI_TUNE=1 python tools/pair_probe.py [channels] [pair:cus:main ...]"""
import os
import sys
import time

os.environ.setdefault("QI_TUNE", "1")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import synth  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
settings = sys.argv[2:] or ["6:224:0", "6:224:1", "6:192:0", "6:192:1", "6:240:0", "7:224:0", "7:192:1"]
n, fs, order = 1 << 20, 1000.0, 3
nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
sig = torch.from_numpy(synth.channels(n, fs, 0, C, C, np.float32)).cuda()
ref = None
rc = 0
for s in ["0:0:0"] + settings + ["0:0:0"]:
    pair, cus, main = s.split(":")
    os.environ["QI_NATIVE_PAIR"], os.environ["QI_NATIVE_PAIR_CUS"], os.environ["QI_NATIVE_PAIR_MAIN"] = pair, cus, main
    plan = qi.TfrPlan(n, torch.float32, "cuda:0", qi.TfrPlan.workspace_for(n, nb, torch.float32, C))
    plan.set_styx_bank(order, fs)
    plan.set_stx_bands(order, fs)
    out = plan.cwt_stx(sig, coef=True, reductions=True)
    for _ in range(300):
        plan.cwt_stx(sig, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            plan.cwt_stx(sig, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    got = [t.clone() for r in out for t in (r.coef, r.reduced)]
    if ref is None:
        ref = got
    same = all(torch.equal(a, b) for a, b in zip(ref, got))
    rc |= 0 if same else 1
    print(f"pair {pair} block CUs {cus} main {main}: {best * 1e3:.4f} ms per step = {2 * C * nb * n / best / 1e6:.0f} Mpoints/s, bit-equal to serial: {same}", flush=True)
    plan.close()
sys.exit(rc)
