"""One-off stress (GPU box): float64 engines (float64 zoom + two-pass + short-atom sub-table) against the hipFFT engine at
2^20 samples over orders the test-suite does not visit -- every row to its own maximum and the fused reductions."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_inferno_amd import engine, _lib, scales_dyadic, synth

bad = 0
n, fs = 1 << 20, 800.0
for order, C in ((1, 1), (2, 2), (4, 1), (8, 3), (24, 1)):
    nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
    rng = np.random.default_rng(order)
    x = np.stack([synth.log_chirp(n, fs, c, C, np.float64) for c in range(C)]) + 0.2 * rng.standard_normal((C, n))
    x = torch.from_numpy(x).cuda()
    nat = engine.TfrPlan(n, torch.float64, None, engine.TfrPlan.workspace_for(n, nb, torch.float64, C), _lib.QI_ENGINE_AUTO)
    ref = engine.TfrPlan(n, torch.float64, None, engine.TfrPlan.workspace_for(n, nb, torch.float64, 1), _lib.QI_ENGINE_HIPFFT)
    for p in (nat, ref):
        p.set_styx_bank(order, fs)
        p.set_stx_bands(order, fs)
    line = f"order {order:2d} C {C}: bands {nb}, zoom {nat.stage_bands('zoom')}, two-pass {nat.stage_bands('pass2')}"
    for name in ("cwt", "stx"):
        a = getattr(nat, name)(x, coef=True, reductions=True)
        c = C - 1
        b = getattr(ref, name)(x[c : c + 1], coef=True, reductions=True)
        peak = b.coef[0].abs().amax(dim=1)
        err = (a.coef[c] - b.coef[0]).abs().amax(dim=1) / peak
        pb = float(((a.power_band[c] - b.power_band[0]).abs() / b.power_band[0]).max())
        st = float(((a.stats[c, :3] - b.stats[0, :3]).abs() / b.stats[0, :3].abs()).max())
        pt = float((a.power_time[c] - b.power_time[0]).abs().max() / b.power_time[0].max())
        ok = float(err.max()) <= 5e-9 and pb <= 1e-9 and st <= 1e-9 and pt <= 1e-10
        bad += 0 if ok else 1
        line += f" | {name}: row {float(err.max()):.1e} band {pb:.1e} stats {st:.1e} time {pt:.1e}{'' if ok else ' FAIL'}"
        del a, b
    print(line, flush=True)
    nat.close(); ref.close(); del x
    torch.cuda.empty_cache()
print("failures:", bad)
sys.exit(1 if bad else 0)
