"""Debug: native engine single-record vs batch reductions, which bands differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tfr_oracle as orc
from quantum_inferno_amd import engine

n, fs, order = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 1000.0, 3
BITS = len(sys.argv) > 2
x = torch.from_numpy(np.stack([orc.synth_chirp(n, fs, c, 3, np.float32) for c in range(3)])).cuda()
from quantum_inferno_amd import scales_dyadic
B = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
plan = engine.TfrPlan(n, np.float32, None, engine.TfrPlan.workspace_for(n, B, np.float32, 3))
plan.set_styx_bank(order, fs)
plan.set_stx_bands(order, fs)
for name in ("cwt", "stx"):
    batch = getattr(plan, name)(x, coef=True, bits=BITS, reductions=True)
    batch2 = getattr(plan, name)(x, coef=True, bits=BITS, reductions=True)
    print(name, "repeat equal:", torch.equal(batch.power_band, batch2.power_band), torch.equal(batch.stats, batch2.stats),
          torch.equal(batch.power_time, batch2.power_time))
    for c in range(3):
        one = getattr(plan, name)(x[c:c + 1], coef=True, reductions=True)
        d = (one.power_band[0] != batch.power_band[c]).nonzero().flatten().tolist()
        print(name, c, "bands differing:", d, "coef equal", torch.equal(one.coef[0], batch.coef[c]))
        if d:
            j = d[0]
            direct = (batch.coef[c, j].abs().double() ** 2).sum().item()
            print("   band", j, one.power_band[0, j].item(), batch.power_band[c, j].item(), "direct", direct)
    lean = getattr(plan, name)(x, coef=False, reductions=True)
    d = (lean.power_band[0] != batch.power_band[0]).nonzero().flatten().tolist()
    print(name, "lean vs full bands differing:", d, "time equal", torch.equal(lean.power_time, batch.power_time))
    if d:
        j = d[0]
        print("   band", j, lean.power_band[0, j].item(), batch.power_band[0, j].item())
