#!/usr/bin/env python3
"""Where the drop-in call's result path spends its time: pinned allocation (fresh / cached), the
device-to-host copy of one widened order-3 panel (805 MB), in one piece and in tiles, and the wrappers themselves.
usage (GPU box): python tools/d2h_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_inferno_amd import engine, styx_cwt, styx_stx, synth  # noqa: E402


def t(fn, sync=True):
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - a) * 1e3, r


nbytes = 48 * (1 << 20) * 16
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
ms, host = t(lambda: torch.empty(nbytes, dtype=torch.uint8, pin_memory=True))
print(f"pinned alloc fresh {nbytes/1e6:.0f} MB: {ms:.1f} ms")
for k in range(3):
    ms, _ = t(lambda: host.copy_(dev))
    print(f"  copy_ whole: {ms:.2f} ms = {nbytes/ms/1e6:.1f} GB/s")
for k in range(2):
    ms, _ = t(lambda: host.copy_(dev, non_blocking=True))
    print(f"  copy_ non_blocking + sync: {ms:.2f} ms = {nbytes/ms/1e6:.1f} GB/s")
tile = nbytes // 12
ms, _ = t(lambda: [host[i * tile:(i + 1) * tile].copy_(dev[i * tile:(i + 1) * tile], non_blocking=True) for i in range(12)])
print(f"  12 tiles non_blocking: {ms:.2f} ms = {nbytes/ms/1e6:.1f} GB/s")
del host
ms, host = t(lambda: torch.empty(nbytes, dtype=torch.uint8, pin_memory=True))
print(f"pinned alloc cached: {ms:.2f} ms")
ms, pg = t(lambda: np.empty(nbytes, dtype=np.uint8))
ms2, _ = t(lambda: pg.fill(1))
print(f"pageable np.empty {ms:.2f} ms, first touch {ms2:.1f} ms")
ms, _ = t(lambda: np.copyto(pg, host.numpy()))
print(f"host memcpy pinned -> pageable (touched): {ms:.1f} ms = {nbytes/ms/1e6:.1f} GB/s")
ms, _ = t(lambda: torch.from_numpy(pg).copy_(dev))
print(f"copy_ device -> pageable: {ms:.1f} ms = {nbytes/ms/1e6:.1f} GB/s")
del host, pg, dev

n, fs, order = 1 << 20, 1000.0, 3
x = synth.channels(n, fs, 0, 1, 1, np.float32)[0]
for mode in ("reference", "native"):
    engine.NUMPY_RESULT_DTYPE = mode
    for k in range(4):
        ms_c, c = t(lambda: styx_cwt.cwt_complex_any_scale_pow2(order, x, fs)[2])
        ms_s, s = t(lambda: styx_stx.stx_complex_any_scale_pow2(order, x, fs)[2])
        print(f"wrappers {mode} call {k}: cwt {ms_c:.1f} ms, stx {ms_s:.1f} ms, pair {ms_c + ms_s:.1f} ms "
              f"= {(c.nbytes + s.nbytes) / (ms_c + ms_s) / 1e6:.1f} GB/s ({c.dtype})")
        del c, s  # a caller that consumes a result and drops it: the page-locked block goes back to the cache
