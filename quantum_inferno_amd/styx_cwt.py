"""
Gabor-atom continuous wavelet transform on the GPU, behind the reference's signatures
(quantum_inferno/styx_cwt.py).  The atom bank is built on the device in float64 exactly as
the reference builds it in the time domain, transformed once per plan, and the zero-padded
linear correlation (scipy.signal.fftconvolve, mode="same") is run as multiply + inverse FFT.
"""
from typing import Tuple, Union

import numpy as np
import torch

from . import _lib, engine
from . import scales_dyadic as scales


def wavelet_variance_theory(amp: float, time_s: np.ndarray, scale: float, omega: float) -> Tuple[float, float]:
    """Nominal variance of the real and imaginary parts of a Gabor atom (ref styx_cwt.py:14-26)."""
    base = amp ** 2 / len(time_s) * 0.5 * np.sqrt(np.pi) * scale
    damp = np.exp(-((scale * omega) ** 2))
    return base / (1 + damp), base / (1 - damp)


def wavelet_amplitude(scale_atom):
    """(canonical unit-norm amplitude, unit-peak-spectrum amplitude) (ref styx_cwt.py:29-40)."""
    amp_canonical = (np.pi * scale_atom ** 2) ** (-1 / 4)
    amp_unit_spectrum = (4 * np.pi * scale_atom ** 2) ** (-1 / 4) * amp_canonical
    return amp_canonical, amp_unit_spectrum


def amplitude_convert_norm_to_spect(scale_atom):
    """Ratio spect / norm amplitude (ref styx_cwt.py:43-55)."""
    a_norm, a_spect = wavelet_amplitude(scale_atom)
    return a_spect / a_norm


def wavelet_time(time_s: np.ndarray, offset_time_s: float, frequency_sample_rate_hz: float) -> np.ndarray:
    """Non-dimensional shifted time fs (t - t0) (ref styx_cwt.py:58-65)."""
    return frequency_sample_rate_hz * (time_s - offset_time_s)


def _tile_like_reference(per_band, n):
    """The reference returns per-band quantities tiled to [B x n] when f is a vector (styx_cwt.py:101-103);
    a broadcast view has the same shape and values without the copy."""
    return np.broadcast_to(np.asarray(per_band)[:, None], (len(per_band), n))


def wavelet_centered_4cwt(
    band_order_nth: float,
    duration_points: int,
    scale_frequency_center_hz: Union[np.ndarray, float],
    frequency_sample_rate_hz: float,
    dictionary_type: str = "norm",
):
    """Gabor atoms centred on the record, evaluated on the GPU in float64 (ref styx_cwt.py:113-144).

    :return: atoms [B x n] complex128 (1-D for a scalar frequency), centred time in s, scale, omega, amp
    """
    scalar = np.isscalar(scale_frequency_center_hz)
    f_hz = np.atleast_1d(np.asarray(scale_frequency_center_hz, dtype=np.float64))
    n = int(duration_points)
    scale, omega = scales.scale_from_frequency_hz(band_order_nth, f_hz, frequency_sample_rate_hz)
    a_norm, a_spect = wavelet_amplitude(scale)
    amp = a_spect if dictionary_type == "spect" else (np.ones(scale.shape) if dictionary_type == "unit" else a_norm)
    atoms = engine.gabor_atoms(n, 0.5 / scale ** 2, np.zeros_like(scale), omega, amp).cpu().numpy()
    time_s = np.arange(n) / frequency_sample_rate_hz
    t_centered = wavelet_time(time_s, time_s[-1] / 2.0, frequency_sample_rate_hz) / frequency_sample_rate_hz
    if scalar:
        return atoms[0], t_centered, scale[0], omega[0], amp[0]
    return atoms, t_centered, _tile_like_reference(scale, n), _tile_like_reference(omega, n), _tile_like_reference(amp, n)


def wavelet_complex(
    band_order_nth: float,
    time_s: np.ndarray,
    offset_time_s: float,
    scale_frequency_center_hz: Union[np.ndarray, float],
    frequency_sample_rate_hz: float,
):
    """Unit-amplitude Gabor atom(s) on an arbitrary time axis and offset (ref styx_cwt.py:68-110): evaluated on the
    GPU at x = fs * (time_s - offset_time_s) whatever the axis is.

    :return: wavelet, shifted time, omega, scale, omega, amp_canonical, amp_unit_spectrum
    """
    time_s = np.asarray(time_s, dtype=np.float64)
    n = len(time_s)
    x = wavelet_time(time_s, offset_time_s, frequency_sample_rate_hz)
    scalar = np.isscalar(scale_frequency_center_hz)
    f_hz = np.atleast_1d(np.asarray(scale_frequency_center_hz, dtype=np.float64))
    scale, omega = scales.scale_from_frequency_hz(band_order_nth, f_hz, frequency_sample_rate_hz)
    atoms = engine.gabor_atoms(n, 0.5 / scale ** 2, np.zeros_like(scale), omega, np.ones_like(scale), x=x).cpu().numpy()
    a_norm, a_spect = wavelet_amplitude(scale)
    if scalar:
        return atoms[0], x, omega[0], scale[0], omega[0], a_norm[0], a_spect[0]
    tile = lambda v: _tile_like_reference(v, n)  # noqa: E731
    return atoms, x, omega, tile(scale), tile(omega), tile(a_norm), tile(a_spect)


def cwt_complex_any_scale_pow2(
    band_order_nth: float,
    sig_wf,
    frequency_sample_rate_hz: float,
    cwt_type: str = "fft",
    dictionary_type: str = "norm",
):
    """Order-N Gabor CWT of one record, or of every row of a [channels x n] batch (ref styx_cwt.py:147-198).

    NumPy in -> NumPy out; CUDA tensor in -> CUDA tensors out.  float64 input is computed in
    float64 / complex128 as the reference does, float32 input in float32 and -- for NumPy callers -- widened to
    the reference's complex128 on the way out (engine.NUMPY_RESULT_DTYPE).

    :return: frequency_cwt_hz [B], time_cwt_s [n], cwt [B x n] (or [channels x B x n])
    """
    if cwt_type != "fft":
        # the reference's "morlet2" branch calls scipy.signal.cwt, removed from SciPy >= 1.15 which it requires
        raise ValueError(f"cwt_type {cwt_type!r} is not available; use 'fft'")
    sig, was_numpy, was_1d = engine.as_signal(sig_wf)
    n = sig.shape[1]
    fs = float(frequency_sample_rate_hz)
    key = ("styx_cwt", n, fs, float(band_order_nth), dictionary_type, sig.dtype, sig.device.index)

    def make():
        f_hz = scales.log_frequency_hz_from_fft_points(fs, n, band_order_nth)
        plan = engine.TfrPlan(n, sig.dtype, sig.device, engine.TfrPlan.workspace_for(n, len(f_hz), sig.dtype))
        plan.set_styx_bank(band_order_nth, fs, dictionary_type)
        return plan

    plan = engine.cached_plan(key, make)
    res = plan.cwt(sig, coef=True)
    time_cwt_s = np.arange(n) / fs
    return res.frequency_hz, time_cwt_s, engine.finish(res.coef, was_numpy, was_1d, widen=True)
