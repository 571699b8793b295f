"""Diagnostic (GPU box): error of the HIP path against the reference fixtures at 2^20 samples, per band and by
bits floor -- the numbers the stated tolerances of tests/test_gpu_parity.py are chosen from."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantum_inferno_amd as qi  # noqa: E402
from quantum_inferno_amd import engine, synth  # noqa: E402

EPS = 2.0 ** -52


def main():
    n, fs = 1 << 20, 1000.0
    for order, name in ((3, "large_n1048576.npz"), (12, "large_n1048576_o12.npz")):
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        x = torch.from_numpy(synth.log_chirp(n, fs, 0, 1, np.float32)).cuda().unsqueeze(0)
        f = qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
        plan = engine.TfrPlan(n, torch.float32, None, engine.TfrPlan.workspace_for(n, len(f), torch.float32, 1))
        plan.set_styx_bank(order, fs)
        plan.set_stx_bands(order, fs)
        for label, res in zip(("cwt", "stx"), plan.cwt_stx(x, coef=True, bits=True, reductions=True)):
            tsel = torch.from_numpy(g[f"{label}_tsel_o{order}"]).cuda()
            got = res.coef[0][:, tsel].cpu().numpy().astype(np.complex128)
            bits = res.bits[0][:, tsel].cpu().numpy().astype(np.float64)
            ref = g[f"{label}_rows_o{order}"]
            err = np.abs(got - ref)
            pmax = np.sqrt(float(g[f"{label}_pmax_o{order}"]))
            rowmax = np.abs(ref).max(axis=1)
            rel_row = err.max(axis=1) / rowmax
            worst = int(np.argmax(rel_row))
            print(f"order {order} {label}: panel-rel {err.max() / pmax:.2e}; worst row-rel {rel_row.max():.2e} (band {worst}, "
                  f"row max / panel max {rowmax[worst] / pmax:.2e}); median row-rel {np.median(rel_row):.2e}")
            print("   row-rel by band:", " ".join(f"{v:.0e}" for v in rel_row))
            ref_bits = np.log2(np.abs(ref) + EPS)
            for floor in (1e-4, 3e-4, 1e-3, 3e-3, 1e-2):
                sel = np.abs(ref) >= floor * pmax
                print(f"   bits: floor {floor:.0e} of the panel max -> max |delta| {np.abs(bits - ref_bits)[sel].max():.2e} over {sel.sum()} samples")
            pb = res.power_band[0].cpu().numpy()
            ref_pb = g[f"{label}_psum_band_o{order}"]
            print(f"   per-band power: max rel to each band {np.max(np.abs(pb - ref_pb) / ref_pb):.2e}, to the largest {np.max(np.abs(pb - ref_pb)) / ref_pb.max():.2e}")
        plan.close()


if __name__ == "__main__":
    main()
