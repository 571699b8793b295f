"""
Order-N constant-Q band tables (host side, NumPy float64).

Which bands exist and which FFT bin each one maps to is integer / float64 work that must
agree with the reference to the last bit, so it stays on the host and is handed to the GPU
plan as tables (include/qi_tfr.h: qi_plan_set_gabor_bank, qi_plan_set_stx_bands).

Mirrors quantum_inferno/scales_dyadic.py of the reference (same names, same argument
meaning); soft failures warn through `warnings` instead of print().
"""
import sys
import warnings
from typing import Tuple, Union

import numpy as np

EPSILON64 = np.finfo(np.float64).eps  # ref scales_dyadic.py:16-18
EPSILON32 = np.finfo(np.float32).eps
EPSILON16 = np.finfo(np.float16).eps

M_OVER_N = 0.75 * np.pi  # cycles per order, ref scales_dyadic.py:21


def get_epsilon() -> float:
    """Machine epsilon matching the interpreter's word size (ref scales_dyadic.py:28-37)."""
    if sys.maxsize > 2 ** 32:
        return EPSILON64
    return EPSILON32 if sys.maxsize > 2 ** 16 else EPSILON16


class Slice:
    """Preferred orders, bases and reference scales (subset of ref scales_dyadic.py:40-81)."""

    ORD1, ORD3, ORD6, ORD12, ORD24, ORD48 = 1.0, 3.0, 6.0, 12.0, 24.0, 48.0
    G2 = 2.0
    G3 = 10.0 ** 0.3
    T_PLANCK = 5.4e-44
    T0S = 1e-42
    T1S = 1.0
    T100S = 100.0
    T1000S = 1000.0
    T1M = 60.0
    T1H = T1M * 60.0
    T1D = T1H * 24.0
    TU = 2.0 ** 58
    F1HZ = 1.0
    F1KHZ = 1_000.0
    F0HZ = 1.0e42
    FU = 2.0 ** -58
    FS1HZ, FS10HZ, FS30HZ, FS80HZ = 1.0, 10.0, 30.0, 80.0
    FS200HZ, FS400HZ, FS800HZ = 200.0, 400.0, 800.0
    FS8KHZ, FS16KHZ, FS48KHZ = 8_000.0, 16_000.0, 48_000.0


DEFAULT_SCALE_BASE = Slice.G3
DEFAULT_SCALE_ORDER = Slice.ORD3
DEFAULT_REF_FREQUENCY_HZ = Slice.F1HZ
DEFAULT_SCALE_ORDER_MIN: float = 0.75
_eps = get_epsilon()
DEFAULT_FFT_POW2_POINTS_MAX: int = 2 ** (63 if _eps == EPSILON64 else (31 if _eps == EPSILON32 else 15))
DEFAULT_FFT_POW2_POINTS_MIN: int = 2 ** 8
DEFAULT_MESH_POW2_PIXELS: int = 2 ** 19
DEFAULT_TIME_DISPLAY_S: float = 60.0
VALID_SCALE_ORDERS = [0.75, 1, 1.5, 3, 6, 12, 24, 48]


def scale_order_check(scale_order: float = DEFAULT_SCALE_ORDER, show_warning: bool = True) -> float:
    """|N|, clamped from below to 0.75 (ref scales_dyadic.py:105-122)."""
    scale_order = np.abs(scale_order)
    if scale_order < DEFAULT_SCALE_ORDER_MIN:
        if show_warning:
            warnings.warn(f"scale order < {DEFAULT_SCALE_ORDER_MIN}: using N = {DEFAULT_SCALE_ORDER_MIN}", stacklevel=2)
        scale_order = DEFAULT_SCALE_ORDER_MIN
    return scale_order


def scale_multiplier(scale_order: float = DEFAULT_SCALE_ORDER) -> float:
    """M = 0.75 pi N (ref scales_dyadic.py:125-130)."""
    return M_OVER_N * scale_order_check(scale_order)


def cycles_from_order(scale_order: float) -> float:
    """Number of cycles M of the order-N atom (ref scales_dyadic.py:133-141)."""
    return scale_multiplier(scale_order)


def order_from_cycles(cycles_per_scale: float) -> float:
    """Inverse of cycles_from_order, at least one cycle (ref scales_dyadic.py:144-155)."""
    if np.abs(cycles_per_scale) < 1:
        cycles_per_scale = 1.0
    return scale_order_check(cycles_per_scale / M_OVER_N)


def base_multiplier(scale_order: float = DEFAULT_SCALE_ORDER, scale_base: float = DEFAULT_SCALE_BASE) -> float:
    """N / log2(G) (ref scales_dyadic.py:158-164)."""
    return scale_order_check(scale_order) / np.log2(scale_base)


def scale_from_frequency_hz(
    scale_order: float, scale_frequency_center_hz: Union[np.ndarray, float], frequency_sample_rate_hz: float
) -> Tuple[Union[np.ndarray, float], Union[np.ndarray, float]]:
    """(atom scale s = M / omega, omega = 2 pi f / fs) (ref scales_dyadic.py:167-180)."""
    omega = 2.0 * np.pi * scale_frequency_center_hz / frequency_sample_rate_hz
    return cycles_from_order(scale_order) / omega, omega


def band_intervals_periods(
    scale_order_input: float,
    scale_base_input: float,
    scale_ref_input: float,
    scale_low_input: float,
    scale_high_input: float,
    show_warnings: bool = True,
):
    """Band numbers, centres and edges in period units (ref scales_dyadic.py:241-352).

    :return: order, base, band_number, ref, centre_algebraic, centre_geometric, start, end
    """

    def note(msg):
        if show_warnings:
            warnings.warn(msg, stacklevel=3)

    ref, low, high, base, order = np.absolute(
        [scale_ref_input, scale_low_input, scale_high_input, scale_base_input, scale_order_input]
    )
    if not (base == Slice.G3 or base == Slice.G2):
        if base < 1.0:
            note("base must exceed one: using G = 2")
            base = Slice.G2
        else:
            note(f"base {base} is neither G2 nor G3")
    if order not in VALID_SCALE_ORDERS:
        if order < 0.75:
            note("order must exceed 0.75: using N = 1")
            order = 1
        else:
            note(f"non-standard order {order}; recommended {VALID_SCALE_ORDERS}")

    edge = base ** (1.0 / (2.0 * order))
    width = edge - 1.0 / edge
    if low < Slice.T0S:
        low = Slice.T0S / edge
    if high < low:
        note("upper scale below the lowest scale: using min = max / G")
        low = high / base
    if high == low:
        note("upper scale equals lowest scale: returning the closest band edges")
        high *= edge
        low /= edge

    n_max = np.round(order * np.log(high / ref) / np.log(base))
    n_min = np.floor(order * np.log(low / ref) / np.log(base))
    # keep the shortest band's lower edge at or above the Nyquist period
    centre_n_min = ref * np.power(base, n_min / order)
    if (centre_n_min < low) or (centre_n_min / edge < low - get_epsilon()):
        n_min += 1
    if n_max < n_min:
        note(f"insufficient bandwidth for order {order} (minimum scaled bandwidth {width}): applying one order")
        n_max = np.floor(np.log10(high) / np.log10(base))
        n_min = n_max - order

    band_number = np.arange(n_min, n_max + 1)
    centre_geometric = ref * np.power(base * np.ones(band_number.shape), band_number / order)
    start = centre_geometric / edge
    end = centre_geometric * edge
    return order, base, band_number, ref, (start + end) / 2.0, centre_geometric, start, end


def band_frequency_low_high(
    frequency_order_input: float,
    frequency_base_input: float,
    frequency_ref_input: float,
    frequency_low_input: float,
    frequency_high_input: float,
    frequency_sample_rate_input: float,
):
    """Same table in Hz (ref scales_dyadic.py:183-238).

    :return: order, base, band_number, ref, centre_algebraic, centre_geometric, start, end (all Hz)
    """
    period_low = 1 / frequency_high_input
    period_nyquist = 2 / frequency_sample_rate_input
    if period_low < period_nyquist:
        period_low = period_nyquist
    order, base, number, period_ref, _, period_geo, period_start, period_end = band_intervals_periods(
        frequency_order_input, frequency_base_input, 1 / frequency_ref_input, period_low, 1 / frequency_low_input
    )
    f_end = 1 / period_start
    f_start = 1 / period_end
    return order, base, -number, 1 / period_ref, (f_end + f_start) / 2.0, 1 / period_geo, f_start, f_end


def log_frequency_hz_from_fft_points(
    frequency_sample_hz: float,
    fft_points: int,
    scale_order: float = DEFAULT_SCALE_BASE,
    scale_ref_hz: float = DEFAULT_REF_FREQUENCY_HZ,
    scale_base: float = DEFAULT_SCALE_BASE,
) -> np.ndarray:
    """Ascending band centre frequencies supported by an fft_points record, from one band
    below 0.8 Nyquist down to the band whose M-cycle atom still fits (ref scales_dyadic.py:355-393;
    the odd default of scale_order is the reference's and callers always pass N)."""
    log2_points = int(np.ceil(np.log2(fft_points)))
    order_over_log2base = base_multiplier(scale_order, scale_base)
    log2_cycles = np.log2(scale_multiplier(scale_order))
    log2_rate = np.log2(frequency_sample_hz / scale_ref_hz)
    band_first = int(np.ceil(order_over_log2base * (np.log2(2.5) - log2_rate)))
    band_last = int(np.floor(order_over_log2base * (log2_points - log2_cycles - log2_rate)))
    bands = np.arange(band_first, band_last + 1)
    return np.flip(scale_ref_hz * scale_base ** (-bands / scale_order))


def stx_shift_indices(frequency_stx_hz: np.ndarray, fft_points: int, frequency_sample_rate_hz: float) -> np.ndarray:
    """FFT bin nearest to each band centre, first occurrence on ties: the integer selection
    inside the Stockwell loop (ref styx_stx.py:216,233)."""
    grid = np.fft.fftfreq(fft_points, 1 / frequency_sample_rate_hz)
    return np.array([np.abs(grid - f).argmin() for f in np.atleast_1d(frequency_stx_hz)], dtype=np.int64)
