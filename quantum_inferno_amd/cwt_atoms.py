"""
Chirp / Gabor atom CWT on the GPU behind the reference's signatures (quantum_inferno/cwt_atoms.py).
Band tables come from scales_dyadic.band_frequency_low_high with base G2; the "fft" back-end is
the reference's circular correlation of length n followed by a half-record roll, the "conv"
back-end its zero-padded linear correlation.  Both run as multiply + inverse FFT in libqi_tfr.so.
"""
from typing import Tuple, Union

import numpy as np

from . import _lib, engine
from . import scales_dyadic as scales


def chirp_mqg_from_n(band_order_nth: float, index_shift: float = 0, scale_base: float = scales.Slice.G2):
    """(cycles M_q, quality factor Q, gamma) of order N; N < 0.7 reverts to 3 (ref cwt_atoms.py:122-144)."""
    if band_order_nth < 0.7:
        band_order_nth = 3.0
    band_edge = scale_base ** (1.0 / 2.0 / band_order_nth)
    quality_q = 1.0 / (band_edge - 1.0 / band_edge)
    gamma = np.sqrt(np.log(2)) * (1 - np.log(2) * (index_shift / np.pi) ** 2) ** (-0.5)
    return 2 * quality_q * gamma, quality_q, gamma


def chirp_scale(cycles_m: float, scale_frequency_center_hz, frequency_sample_rate_hz: float):
    """Atom scale s = M fs / (2 pi f) (ref cwt_atoms.py:147-158)."""
    return cycles_m * frequency_sample_rate_hz / scale_frequency_center_hz / (2.0 * np.pi)


def chirp_p_complex(scale_atom, gamma: float, index_shift: float):
    """p = (1 - i shift gamma / pi) / (2 s^2) (ref cwt_atoms.py:202-211)."""
    return (1 - 1j * index_shift * gamma / np.pi) / (2 * scale_atom ** 2)


def chirp_amplitude(scale_atom, gamma: float, index_shift: float):
    """(unit-norm amplitude, unit-spectrum amplitude) (ref cwt_atoms.py:214-226)."""
    p = chirp_p_complex(scale_atom, gamma, index_shift)
    return 1 / np.pi ** 0.25 * 1 / np.sqrt(scale_atom), np.sqrt(np.abs(p) / np.pi)


def chirp_time(time_s: np.ndarray, offset_time_s: float, frequency_sample_rate_hz: float) -> np.ndarray:
    """fs (t - t0) (ref cwt_atoms.py:229-238)."""
    return frequency_sample_rate_hz * (time_s - offset_time_s)


def chirp_scales_from_duration(
    band_order_nth: float, sig_duration_s: float, index_shift: float = 0.0, scale_base: float = scales.Slice.G2
) -> Tuple[float, float]:
    """(duration / M, M / duration): longest supported atom time scale and its frequency (ref cwt_atoms.py:241-256)."""
    cycles_m, _, _ = chirp_mqg_from_n(band_order_nth, index_shift, scale_base)
    scale_time_s = sig_duration_s / cycles_m
    return scale_time_s, 1 / scale_time_s


def chirp_frequency_bands(
    scale_order_input: float,
    frequency_low_input: float,
    frequency_sample_rate_input: float,
    frequency_high_input: float,
    index_shift: float = 0,
    frequency_ref: float = scales.Slice.F1HZ,
    scale_base: float = scales.Slice.G2,
):
    """Band centres / edges between two frequencies, in descending frequency as the reference
    returns them (ref cwt_atoms.py:259-300).

    :return: order, cycles M, Q, gamma, centre_geometric, start, end
    """
    order, base, _, _, _, f_geo, f_start, f_end = scales.band_frequency_low_high(
        scale_order_input, scale_base, frequency_ref, frequency_low_input, frequency_high_input,
        frequency_sample_rate_input,
    )
    cycles_m, quality_q, gamma = chirp_mqg_from_n(order, index_shift, base)
    return order, cycles_m, quality_q, gamma, f_geo, f_start, f_end


def _atom_tables(order, f_hz, fs, index_shift, scale_base, dictionary_type):
    cycles_m, _, gamma = chirp_mqg_from_n(order, index_shift, scale_base)
    scale = chirp_scale(cycles_m, np.asarray(f_hz, dtype=np.float64), fs)
    p = chirp_p_complex(scale, gamma, index_shift)
    a_norm, a_spect = chirp_amplitude(scale, gamma, index_shift)
    amp = a_norm if dictionary_type == "norm" else a_spect
    return np.real(p), np.imag(p), cycles_m / scale, amp * np.ones_like(scale)


def chirp_centered_4cwt(
    band_order_nth: float,
    sig_or_time: np.ndarray,
    scale_frequency_center_hz: float,
    frequency_sample_rate_hz: float,
    index_shift: float = 0,
    scale_base: float = scales.Slice.G2,
    dictionary_type: str = "norm",
):
    """One atom centred on a record of len(sig_or_time) points, evaluated on the GPU in float64
    (ref cwt_atoms.py:303-340).  :return: atom [n] complex128, centred time in s"""
    n = len(sig_or_time)
    fs = frequency_sample_rate_hz
    tabs = _atom_tables(band_order_nth, [scale_frequency_center_hz], fs, index_shift, scale_base, dictionary_type)
    atom = engine.gabor_atoms(n, *tabs).cpu().numpy()[0]
    time_s = np.arange(n) / fs
    return atom, chirp_time(time_s, time_s[-1] / 2.0, fs) / fs


def chirp_complex(
    band_order_nth: float,
    time_s: np.ndarray,
    offset_time_s: float,
    scale_frequency_center_hz: float,
    frequency_sample_rate_hz: float,
    index_shift: float = 0,
    scale_base: float = scales.Slice.G2,
):
    """Unit-amplitude chirp atom on any time axis and offset (ref cwt_atoms.py:16-50).
    :return: atom, shifted time in s, normal_scaling, spectrum_scaling"""
    time_s = np.asarray(time_s, dtype=np.float64)
    n = len(time_s)
    fs = frequency_sample_rate_hz
    x = chirp_time(time_s, offset_time_s, fs)
    p_re, p_im, omega, _ = _atom_tables(band_order_nth, [scale_frequency_center_hz], fs, index_shift, scale_base, "norm")
    atom = engine.gabor_atoms(n, p_re, p_im, omega, np.ones(1), x=x).cpu().numpy()[0]
    cycles_m, _, gamma = chirp_mqg_from_n(band_order_nth, index_shift, scale_base)
    a_norm, a_spect = chirp_amplitude(chirp_scale(cycles_m, scale_frequency_center_hz, fs), gamma, index_shift)
    return atom, x / fs, a_norm, a_spect


def cwt_chirp_complex(
    band_order_nth: float,
    sig_wf,
    frequency_low_hz: float,
    frequency_sample_rate_hz: float,
    frequency_high_hz: float = scales.Slice.F0HZ,
    cwt_type: str = "fft",
    index_shift: float = 0,
    frequency_ref: float = scales.Slice.F1HZ,
    scale_base: float = scales.Slice.G2,
    dictionary_type: str = "norm",
):
    """CWT with chirp / Gabor atoms over [frequency_low_hz, frequency_high_hz] (ref cwt_atoms.py:343-444).

    :return: cwt [B x n], cwt_bits [B x n], time_s [n], frequency_cwt_hz [B] ascending
    """
    if cwt_type == "morlet2":
        # dead upstream too: scipy.signal.cwt is gone from the SciPy the reference requires
        raise ValueError("cwt_type 'morlet2' is not available; use 'fft' or 'conv'")
    if cwt_type not in ("fft", "conv"):
        raise ValueError(f"Incorrect cwt_type: {cwt_type} specified in cwt_chirp_complex")
    sig, was_numpy, was_1d = engine.as_signal(sig_wf)
    n = sig.shape[1]
    fs = float(frequency_sample_rate_hz)
    if frequency_high_hz > fs / 2.0:
        frequency_high_hz = fs / 2.0
    key = ("cwt_atoms", n, fs, float(band_order_nth), float(frequency_low_hz), float(frequency_high_hz), cwt_type,
           float(index_shift), float(frequency_ref), float(scale_base), dictionary_type, sig.dtype, sig.device.index)
    which = _lib.QI_BANK_ATOMS if cwt_type == "fft" else _lib.QI_BANK_STYX

    def make():
        order, _, _, _, f_desc, _, _ = chirp_frequency_bands(
            band_order_nth, frequency_low_hz, fs, frequency_high_hz, index_shift, frequency_ref, scale_base
        )
        f_hz = np.flip(f_desc)
        plan = engine.TfrPlan(n, sig.dtype, sig.device, engine.TfrPlan.workspace_for(n, len(f_hz), sig.dtype))
        plan.set_gabor_bank(which, f_hz, *_atom_tables(order, f_hz, fs, index_shift, scale_base, dictionary_type))
        return plan

    plan = engine.cached_plan(key, make)
    res = plan._run(which, sig, True, True, False, 1.0, 0.0)
    return (
        engine.finish(res.coef, was_numpy, was_1d, widen=True),
        engine.finish(res.bits, was_numpy, was_1d, widen=True),
        np.arange(n) / fs,
        res.frequency_hz,
    )


def cwt_chirp_from_sig(
    sig_wf,
    frequency_sample_rate_hz: float,
    band_order_nth: float = 3,
    cwt_type: str = "fft",
    index_shift: float = 0,
    frequency_ref: float = scales.Slice.F1HZ,
    scale_base: float = scales.Slice.G2,
    dictionary_type: str = "norm",
):
    """CWT from the lowest frequency the record supports up to Nyquist (ref cwt_atoms.py:447-486).

    :return: cwt, cwt_bits, time_s, frequency_cwt_hz
    """
    n = sig_wf.shape[-1] if hasattr(sig_wf, "shape") else len(sig_wf)
    _, f_min = chirp_scales_from_duration(band_order_nth, n / frequency_sample_rate_hz, index_shift, scale_base)
    return cwt_chirp_complex(
        band_order_nth, sig_wf, f_min, frequency_sample_rate_hz, frequency_sample_rate_hz / 2.0, cwt_type,
        index_shift, frequency_ref, scale_base, dictionary_type,
    )
