"""
Stockwell transform on the GPU behind the reference's signatures (quantum_inferno/styx_stx.py):
one forward FFT per record, per band a circularly shifted spectrum times a Gaussian that is
regenerated in registers, one inverse FFT per band.
"""
import numpy as np
import torch

from . import _lib, engine
from . import scales_dyadic as scales
from .utilities.rescaling import is_power_of_two


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
    return res.frequency_hz, np.arange(n) / fs, engine.finish(res.coef, was_numpy, was_1d, widen=True)


def sig_pad_up_to_pow2(sig_wf: np.ndarray, n_fft: int, verbosity: bool = False):
    """Zero-pad to n_fft points (ref styx_stx.py:15-48).  Upstream this helper raises TypeError as soon as padding
    is actually needed (tuple + int at :44) and for n_fft=None (comparison at :30); here both cases are handled as
    the docstring upstream describes: None means the next power of two, and the pad is appended."""
    sig_wf = np.asarray(sig_wf)
    n_times = sig_wf.shape[-1]
    if n_fft is None or (not is_power_of_two(int(n_fft)) and n_times > n_fft):
        n_fft = 2 ** int(np.ceil(np.log2(n_times)))
    if n_fft < n_times:
        raise ValueError(f"n_fft cannot be smaller than signal size. Got {n_fft} < {n_times}.")
    zero_pad = int(n_fft - n_times)
    if zero_pad > 0:
        sig_wf = np.concatenate((sig_wf, np.zeros(sig_wf.shape[:-1] + (zero_pad,), sig_wf.dtype)), axis=-1)
    return sig_wf, int(n_fft), zero_pad


def tfr_stx_fft(
    sig_wf: np.ndarray,
    time_sample_interval: float,
    scale_order_input: float = 8.0,
    n_fft_in: int = None,
    frequency_min: float = None,
    frequency_max: float = None,
    frequency_step: float = None,
    factor_q: float = 0.0,
    power_p: float = 0.0,
    power_r: float = 1.0,
    is_geometric: bool = False,
    is_inferno: bool = False,
    scale_base_input: float = scales.Slice.G3,
    scale_ref_input: float = scales.Slice.T1S,
):
    """General Stockwell transform: linear, geometric or inferno-standard bands, sigma tuned by q, p, r
    (ref styx_stx.py:52-192).  Band selection and the shift indices are computed on the host exactly as upstream;
    the transform is the same GPU path as stx_complex_any_scale_pow2 with this band table.

    :return: tfr_stx [B x n], psd_stx = |tfr|^2 + eps [B x n], frequency_stx [B], frequency_stx_fft [B] (the bins the
             bands were snapped to), windows_fft [B x n_fft] complex128 (the Gaussian windows)
    """
    fs: float = 1 / time_sample_interval
    cycles_m: float = 12.0 / 5.0 * scale_order_input
    lin_fft_decimate: float = 2.0
    sig_pow2, n_fft, zero_pad = sig_pad_up_to_pow2(sig_wf, n_fft_in)
    n_out = n_fft - zero_pad
    frequency_fft = np.fft.fftfreq(n_fft, time_sample_interval)
    omega_fft = 2 * np.pi * frequency_fft / fs
    frequency_min_nth = cycles_m / (n_fft / fs)
    if frequency_min is None:
        frequency_min = frequency_min_nth
    if frequency_max is None:
        frequency_max = fs / 2.0
    start_idx = np.abs(frequency_fft - frequency_min).argmin()
    stop_idx = np.abs(frequency_fft - frequency_max).argmin()
    f_start, f_stop = frequency_fft[start_idx], frequency_fft[stop_idx]
    if frequency_step is None:
        frequency_step = (frequency_max - frequency_min) * lin_fft_decimate / len(frequency_fft)
    frequency_stx = np.arange(f_start, f_stop, frequency_step)
    if is_geometric is True:
        if is_inferno is True:
            frequency_stx = scales.band_frequency_low_high(
                scale_order_input, scale_base_input, scale_ref_input, f_start, f_stop, fs
            )[5]
        else:
            num_bands = int(np.log2(f_stop / f_start) * scale_order_input)
            frequency_stx = np.logspace(np.log2(f_start), np.log2(f_stop), num=num_bands, base=scale_base_input)
    n_b = len(frequency_stx)
    if n_b == 0:
        raise ValueError("tfr_stx_fft: the frequency selection is empty")
    idx = np.array([np.abs(frequency_fft - f).argmin() for f in frequency_stx], dtype=np.int64)
    frequency_stx_fft = frequency_fft[idx]
    omega_sx = 2 * np.pi * frequency_stx_fft / fs
    if np.any(omega_sx == 0.0):
        # upstream takes len() of an int here (styx_stx.py:173) and raises TypeError
        raise ValueError("tfr_stx_fft: a band snapped to the zero-frequency bin; raise frequency_min")
    sigma = cycles_m / omega_sx * ((1 + factor_q * (omega_sx ** power_p)) * (omega_sx ** (1 - power_r)))

    sig, was_numpy, was_1d = engine.as_signal(sig_pow2)
    if not was_1d:
        raise ValueError("tfr_stx_fft takes one record")
    plan = engine.TfrPlan(n_fft, sig.dtype, sig.device, engine.TfrPlan.workspace_for(n_fft, n_b, sig.dtype))
    try:
        ia, ip = _lib.iarr(idx % n_fft)
        sa, sp = _lib.darr(sigma)
        _lib.check(plan._lib.qi_plan_set_stx_bands(plan._handle, n_b, ip, sp))
        plan.freq[_lib.QI_TABLE_STX] = frequency_stx
        tfr = plan.stx(sig, coef=True).coef[0, :, :n_out].contiguous()
    finally:
        plan.close()
    # the two by-products the reference also returns: psd = |tfr|^2 + eps and the Gaussian windows on the FFT bins
    # (styx_stx.py:187-192) -- host arithmetic (NumPy), handed back on the device only for tensor callers
    win = np.exp(-0.5 * sigma[:, None] ** 2 * omega_fft[None, :] ** 2).astype(np.complex128)
    if was_numpy:
        tfr_np = engine.finish(tfr, True, False, widen=True)
        psd = tfr_np.real ** 2 + tfr_np.imag ** 2 + float(scales.get_epsilon())
        return tfr_np, psd, frequency_stx, frequency_stx_fft, win
    # (tensor callers: the same by-product as a device expression)
    psd = tfr.real ** 2 + tfr.imag ** 2 + float(scales.get_epsilon())
    return tfr, psd, frequency_stx, frequency_stx_fft, torch.from_numpy(win).to(tfr.device)
