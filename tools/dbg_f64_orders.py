import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantum_inferno_amd import engine, scales_dyadic
for log2n, order in ((16, 1.0), (18, 2.0), (17, 1.0), (20, 1.0)):
    n, fs = 1 << log2n, 1000.0
    nb = len(scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order))
    p = engine.TfrPlan(n, np.float64, None, engine.TfrPlan.workspace_for(n, nb, np.float64, 1))
    p.set_styx_bank(order, fs); p.set_stx_bands(order, fs)
    print('2^%d order %g bands %d zoom %s block %s pass2 %s' % (log2n, order, nb, p.stage_bands('zoom'), p.stage_bands('block'), p.stage_bands('pass2')), flush=True)
    p.close()
