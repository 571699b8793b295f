import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_inferno_amd as qi
from quantum_inferno_amd import synth, _lib
n, fs, order = 1 << 20, 1000.0, 3.0
nb = len(qi.scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
sig = torch.from_numpy(synth.channels(n, fs, 0, 1, 1)).cuda()
res = {}
for eng in (_lib.QI_ENGINE_HIPFFT, _lib.QI_ENGINE_AUTO):
    plan = qi.TfrPlan(n, torch.float32, "cuda:0", qi.TfrPlan.workspace_for(n, nb, torch.float32, 1), eng)
    plan.set_styx_bank(order, fs); plan.set_stx_bands(order, fs)
    res[eng] = (plan.cwt(sig, coef=True).coef[0].clone(), plan.stx(sig, coef=True).coef[0].clone())
    plan.close()
for name, i in (("cwt", 0), ("stx", 1)):
    a, b = res[_lib.QI_ENGINE_HIPFFT][i], res[_lib.QI_ENGINE_AUTO][i]
    d = (a - b).abs(); scale = a.abs().max()
    bad = d > 1e-4 * scale
    print(name, "max rel", float(d.max() / scale), "bad count", int(bad.sum()))
    if bad.any():
        jb = bad.any(dim=1).nonzero().flatten().tolist()
        print("  bad bands", jb)
        j = jb[0]; t = bad[j].nonzero().flatten()
        print("  band", j, "bad t count", len(t), "first", t[:20].tolist(), "mod 16:", sorted(set((t % 16).tolist()))[:16])
        N1 = 1024
        print("   t//N1 (t2) set:", sorted(set(((t // N1)).tolist()))[:40])
