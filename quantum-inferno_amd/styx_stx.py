"""
Stockwell transform on the GPU behind the reference's signatures (quantum_inferno/styx_stx.py):
one forward FFT per record, per band a circularly shifted spectrum times a Gaussian that is
regenerated in registers, one inverse FFT per band.
"""
import numpy as np

from . import engine
from . import scales_dyadic as scales


def stx_complex_any_scale_pow2(band_order_nth: float, sig_wf, frequency_sample_rate_hz: float):
    """Order-N Stockwell transform of one record or of every row of a [channels x n] batch
    (ref styx_stx.py:195-236).  NumPy in -> NumPy out; CUDA tensor in -> CUDA tensors out.

    :return: frequency_stx_hz [B], time_stx_s [n], tfr_stx [B x n] (or [channels x B x n])
    """
    sig, was_numpy, was_1d = engine.as_signal(sig_wf)
    n = sig.shape[1]
    fs = float(frequency_sample_rate_hz)
    key = ("styx_stx", n, fs, float(band_order_nth), sig.dtype, sig.device.index)

    def make():
        f_hz = scales.log_frequency_hz_from_fft_points(fs, n, band_order_nth)
        plan = engine.TfrPlan(n, sig.dtype, sig.device, engine.TfrPlan.workspace_for(n, len(f_hz), sig.dtype))
        plan.set_stx_bands(band_order_nth, fs)
        return plan

    plan = engine.cached_plan(key, make)
    res = plan.stx(sig, coef=True)
    return res.frequency_hz, np.arange(n) / fs, engine.finish(res.coef, was_numpy, was_1d)
