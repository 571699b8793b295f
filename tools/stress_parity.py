"""One-off stress (GPU box): native engines against the hipFFT engine over orders, lengths and record counts that the
test-suite does not visit -- every row to its own maximum.  Prints one line per case; exits 1 on a failure."""
import itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_inferno_amd import engine, _lib, scales_dyadic, synth

bad = 0
import os as _os
cases = [(o, l, c) for o in (1, 2, 4, 8, 24) for l in (15, 17, 19, 20) for c in (1, 5)] + [(3, 22, 1), (12, 22, 1), (5, 21, 4), (1, 22, 1), (2, 21, 2), (1.5, 18, 1), (0.75, 16, 1)]
if _os.environ.get("QI_STRESS_CASES"):
    cases = [tuple(float(v) if "." in v else int(v) for v in c.split(":")) for c in _os.environ["QI_STRESS_CASES"].split(",")]
for order, log2n, C in cases:
    n, fs = 1 << log2n, 800.0
    if order == 24 and log2n > 19:
        continue
    nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
    rng = np.random.default_rng(int(order * 100) + log2n)
    x = np.stack([synth.log_chirp(n, fs, c, C, np.float32) for c in range(C)]) + 0.2 * rng.standard_normal((C, n)).astype(np.float32)
    x = torch.from_numpy(x).cuda()
    nat = engine.TfrPlan(n, torch.float32, None, engine.TfrPlan.workspace_for(n, nb, torch.float32, C), _lib.QI_ENGINE_AUTO)
    ref = engine.TfrPlan(n, torch.float32, None, engine.TfrPlan.workspace_for(n, nb, torch.float32, 1), _lib.QI_ENGINE_HIPFFT)
    for p in (nat, ref):
        p.set_styx_bank(order, fs)
        p.set_stx_bands(order, fs)
    native = [nat.stage_bands("zoom")[w] + nat.stage_bands("block")[w] + (nat.stage_bands("pass2")[w] if log2n == 20 else 0) for w in (0, 2)]
    a_c, a_s = nat.cwt_stx(x, coef=True, reductions=True)
    worst = {}
    for name, a in (("cwt", a_c), ("stx", a_s)):
        for c in sorted({0, C - 1}):
            b = getattr(ref, name)(x[c : c + 1], coef=True, reductions=True)
            peak = b.coef[0].abs().amax(dim=1)
            err = ((a.coef[c] - b.coef[0]).abs().amax(dim=1) / peak)
            pb = float(((a.power_band[c] - b.power_band[0]).abs() / b.power_band[0]).max())
            worst[name] = max(worst.get(name, 0.0), float(err.max()))
            if float(err.max()) > 2e-5 or pb > 1e-4:
                bad += 1
                print(f"  FAIL {name} order {order} n 2^{log2n} C {C} record {c}: row {int(err.argmax())} err {float(err.max()):.2e} band-power {pb:.2e}")
            del b
    print(f"order {order:4g} n 2^{log2n} C {C}: bands {nb}, native bands {native}, worst row-rel cwt {worst['cwt']:.1e} stx {worst['stx']:.1e}", flush=True)
    nat.close(); ref.close(); del a_c, a_s, x
    torch.cuda.empty_cache()
print("failures:", bad)
sys.exit(1 if bad else 0)
