#!/usr/bin/env python3
"""Dispatch timeline of the joint block launch (k_block_dual) of one cwt_stx step, from a -DQI_NATIVE_STAMPS build
(python quantum_inferno_amd/_build.py --variant tl -DQI_NATIVE_STAMPS):
  QI_TFR_LIB=$PWD/quantum_inferno_amd/libqi_tfr_tl.so python tools/block_timeline.py [order] [channels] [out.txt]
Prints the launch's span, the workgroups resident over time, and the durations per item kind."""
import os
import sys

import numpy as np  # noqa: E402


def analyze(out):
    rows = np.loadtxt(out, skiprows=1, dtype=np.int64)
    t0 = rows[:, 1].min()
    start, end = (rows[:, 1] - t0) / 100.0, (rows[:, 2] - t0) / 100.0  # microseconds
    print(f"{len(rows)} workgroups, span {end.max():.1f} us, sum of durations {np.sum(end - start):.0f} us "
          f"= {np.sum(end - start) / end.max():.0f} resident on average")
    slots = len(set(zip(rows[:, 3], rows[:, 4])))
    print(f"{slots} distinct (XCD, CU) places")
    for lo in np.arange(0.0, end.max(), max(5.0, round(end.max() / 20))):
        mid = lo + 0.5
        print(f"  t = {lo:6.1f} us: {int(np.sum((start <= mid) & (end > mid))):5d} resident")
    kinds = sorted(set(zip(rows[:, 6], rows[:, 5])))
    for wq, bands in kinds:
        m = (rows[:, 6] == wq) & (rows[:, 5] == bands)
        d = (end - start)[m]
        print(f"  reach code {wq:3d}, {bands:2d} bands: {m.sum():5d} items, {d.mean():6.1f} us each (min {d.min():.1f}, max {d.max():.1f}), "
              f"started {start[m].min():.1f} .. {start[m].max():.1f} us")


if len(sys.argv) > 2 and sys.argv[1] == "--analyze":  # a timeline file written earlier: no GPU needed
    analyze(sys.argv[2])
    sys.exit(0)
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/block_timeline.txt"
os.environ.update({"QI_TUNE": "1", "QI_NATIVE_STAMPS": "1", "QI_NATIVE_TIMELINE": out})
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import synth  # noqa: E402

order = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
n_ch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, fs = 1 << 20, 1000.0
nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
plan = qi.TfrPlan(n, torch.float32, "cuda:0", qi.TfrPlan.workspace_for(n, nb, torch.float32, n_ch, cap_bytes=64 << 30))
plan.set_styx_bank(order, fs)
plan.set_stx_bands(order, fs)
sig = torch.from_numpy(synth.channels(n, fs, 0, n_ch, n_ch)).cuda()
o = plan.cwt_stx(sig, coef=True, reductions=True)
for _ in range(5):
    plan.cwt_stx(sig, out=o)
torch.cuda.synchronize()
plan.close()
analyze(out)
