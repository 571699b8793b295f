#!/usr/bin/env python3
"""Band tables for the host sanitizer walk (tests/sanitize/walk.cpp): for every (order, record length, precision) the host
arrays the C ABI takes -- qi_plan_set_gabor_bank (styx bank: p_re, p_im, omega, amp), qi_plan_set_stx_bands (shift index,
sigma) -- and the workspace TfrPlan.workspace_for sizes for 1 / 4 / 16 / 64 records, as one little-endian binary file.
Runs in the ordinary interpreter (no sanitizer): the library's own host modules make the tables."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (engine imports it)

from quantum_inferno_amd import engine, scales_dyadic as scales  # noqa: E402

ORDERS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
LOG2N = [14, 15, 16, 17, 18, 19, 20, 21, 22]
RECORDS = [1, 4, 16, 64]


def main(path):
    fs = 1000.0
    with open(path, "wb") as fh:
        fh.write(struct.pack("<4i", 0x51495354, len(ORDERS) * len(LOG2N) * 2, len(RECORDS), 0))
        for order in ORDERS:
            for log2n in LOG2N:
                n = 1 << log2n
                f_hz, p_re, p_im, omega, amp, _ = engine.styx_bank_tables(order, n, fs)
                idx = scales.stx_shift_indices(f_hz, n, fs).astype(np.int64)
                sigma = (scales.cycles_from_order(order) / (2 * np.pi * f_hz / fs)).astype(np.float64)
                for dtype in (0, 1):  # QI_F32, QI_F64
                    td = torch.float64 if dtype else torch.float32
                    ws = [int(engine.TfrPlan.workspace_for(n, len(f_hz), td, c, cap_bytes=48 << 30)) for c in RECORDS]
                    fh.write(struct.pack("<4i", order, log2n, dtype, len(f_hz)))
                    fh.write(struct.pack(f"<{len(RECORDS)}q", *RECORDS))
                    fh.write(struct.pack(f"<{len(RECORDS)}q", *ws))
                    for a in (p_re, p_im, omega, amp, sigma):
                        fh.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
                    fh.write(np.ascontiguousarray(idx, dtype="<i8").tobytes())


if __name__ == "__main__":
    main(sys.argv[1])
