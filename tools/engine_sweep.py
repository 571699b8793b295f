#!/usr/bin/env python3
"""Which engine takes each table: orders x power-of-two lengths x precisions (plan construction only).  Prints the rows of
each table on the zoom / block engines and what is left (two-pass kernels at 2^19 / 2^20, else the hipFFT engine's pass
behind the native run), and flags tables that go to the hipFFT engine as a whole.  tools/engine_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from quantum_inferno_amd import engine, scales_dyadic  # noqa: E402

whole = 0
for dtype in (np.float32, np.float64):
    for order in (1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0):
        for log2n in range(15, 23):
            n, fs = 1 << log2n, 1000.0
            nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
            p = engine.TfrPlan(n, dtype, None, 3 << 30)
            p.set_styx_bank(order, fs)
            p.set_stx_bands(order, fs)
            z, b = p.stage_bands("zoom"), p.stage_bands("block")
            row = []
            for t, name in ((0, "styx"), (2, "stx")):
                nat = z[t] + b[t]
                row.append(f"{name} {nat}/{nb}")
                if nat == 0:
                    whole += 1
                    row[-1] += " WHOLE-TABLE-HIPFFT"
            print(f"{np.dtype(dtype).name} order {order:g} 2^{log2n}: " + ", ".join(row), flush=True)
            p.close()
print("tables on the hipFFT engine as a whole:", whole)
