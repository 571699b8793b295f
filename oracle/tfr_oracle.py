"""
CPU oracle for the TFR hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy (float64 / complex128) restatement of the algorithms of
ISLA-UH/quantum-inferno v1.1.3 for the path named in BASELINE.json:north_star
(styx_fft STFT, styx_cwt CWT + Gabor bank, styx_stx Stockwell, cwt_atoms,
tfr_info).  It exists so that the HIP path can be checked on a box that has
no copy of the reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product package never does.

Pinning: every function below is checked against outputs of the reference
itself (imported from /root/reference in the build container by
oracle/gen_golden.py, which writes tests/golden/*.npz) in
tests/test_oracle_golden.py, and against the one known-answer vector the
reference's own tests hold for this path (tests/test_scales_dyadic.py:8-21,
commented out upstream).  Third-party arithmetic the reference calls
(scipy.signal.stft, scipy.signal.fftconvolve -- SciPy 1.15.3 in the build
container; pyproject.toml:19-20 only says scipy>=1.15.0, numpy>=2.2.1) is
restated here from its published algorithm; the FFT primitive itself is
numpy.fft / scipy.fft (pocketfft).

All `ref:` citations are file:line under /root/reference/quantum_inferno/.
"""
from __future__ import annotations

import numpy as np
import scipy.fft as _sfft
from numpy.fft import fft as _np_fft, ifft as _np_ifft  # cwt_atoms uses numpy.fft (ref: cwt_atoms.py:407,419)
from scipy.fft import fft as _fft, ifft as _ifft, rfft as _rfft  # styx_* use scipy.fft / scipy.signal

EPS64 = np.finfo(np.float64).eps  # np.float64 on purpose (strong NumPy-2 promotion).  ref: scales_dyadic.py:16
EPS32 = np.finfo(np.float32).eps  # ref: scales_dyadic.py:17
G2 = 2.0  # ref: scales_dyadic.py:52
G3 = 10.0 ** 0.3  # ref: scales_dyadic.py:53
ORDER_MIN = 0.75  # ref: scales_dyadic.py:96
M_OVER_N = 0.75 * np.pi  # ref: scales_dyadic.py:21


# --------------------------------------------------------------------------- band tables
def order_check(order):
    """ref: scales_dyadic.py:105-122 (abs, clamp to 0.75)."""
    order = np.abs(order)
    return ORDER_MIN if order < ORDER_MIN else order


def cycles_from_order(order):
    """M = 0.75*pi*N.  ref: scales_dyadic.py:125-141."""
    return M_OVER_N * order_check(order)


def band_table(fs, n_points, order, ref_hz=1.0, base=G3):
    """Band centre frequencies (ascending).  ref: scales_dyadic.py:355-393."""
    log2_len = int(np.ceil(np.log2(n_points)))
    n_over_log2g = order_check(order) / np.log2(base)  # ref: :158-164
    log2_m = np.log2(cycles_from_order(order))
    log2_ref = np.log2(fs / ref_hz)
    band_aa = int(np.ceil(n_over_log2g * (np.log2(2.5) - log2_ref)))
    band_max = int(np.floor(n_over_log2g * (log2_len - log2_m - log2_ref)))
    bands = np.arange(band_aa, band_max + 1)
    return np.flip(ref_hz * base ** (-bands / order))


def scale_omega(order, f_hz, fs):
    """(scale, omega) of the canonical Gabor atom.  ref: scales_dyadic.py:167-180."""
    omega = 2.0 * np.pi * f_hz / fs
    return cycles_from_order(order) / omega, omega


def band_intervals_periods(order_in, base_in, ref_in, low_in, high_in):
    """ISO-style band numbers / centres / edges in period units.
    ref: scales_dyadic.py:241-352 (warnings dropped, arithmetic kept)."""
    ref, low, high, base, order = np.absolute([ref_in, low_in, high_in, base_in, order_in])
    if not (base == G3 or base == G2) and base < 1.0:
        base = G2
    if order not in [0.75, 1, 1.5, 3, 6, 12, 24, 48] and order < 0.75:
        order = 1
    edge = base ** (1.0 / (2.0 * order))
    width = edge - 1.0 / edge
    if low < 1e-42:
        low = 1e-42 / edge
    if high < low:
        low = high / base
    if high == low:
        high *= edge
        low /= edge
    n_max = np.round(order * np.log(high / ref) / np.log(base))
    n_min = np.floor(order * np.log(low / ref) / np.log(base))
    centre_min = ref * np.power(base, n_min / order)
    if (centre_min < low) or (centre_min / edge < low - EPS64):
        n_min += 1
    if n_max < n_min:
        n_max = np.floor(np.log10(high) / np.log10(base))
        n_min = n_max - order
    band_number = np.arange(n_min, n_max + 1)
    centre_geo = ref * np.power(base * np.ones(band_number.shape), band_number / order)
    start = centre_geo / edge
    end = centre_geo * edge
    return order, base, band_number, ref, (start + end) / 2.0, centre_geo, start, end


def band_frequency_low_high(order_in, base_in, ref_hz, f_low, f_high, fs):
    """Same in frequency units.  ref: scales_dyadic.py:183-238."""
    s_low = 1 / f_high
    if s_low < 2 / fs:
        s_low = 2 / fs
    order, base, num, s_ref, _, s_geo, s_start, s_end = band_intervals_periods(
        order_in, base_in, 1 / ref_hz, s_low, 1 / f_low
    )
    f_end = 1 / s_start
    f_start = 1 / s_end
    return order, base, -num, 1 / s_ref, (f_end + f_start) / 2.0, 1 / s_geo, f_start, f_end


# --------------------------------------------------------------------------- small helpers
def to_log2_with_epsilon(x):
    """log2(|x| + eps64).  ref: utilities/rescaling.py:13-20."""
    return np.log2(np.abs(x) + EPS64)


def round_value(value, kind):
    """ref: utilities/calculations.py:160-184."""
    if kind == "floor":
        return int(np.floor(value))
    if kind == "ceil":
        return int(np.ceil(value))
    if kind == "round":
        return int(np.round(value))
    if kind == "ceil_power_of_two":
        return 2 ** int(np.ceil(np.log2(value)))
    if kind == "floor_power_of_two":
        return 2 ** int(np.floor(np.log2(value)))
    raise ValueError(kind)


def get_num_points(fs, duration_s, kind, unit):
    """ref: utilities/calculations.py:187-205."""
    if unit == "points":
        return round_value(fs * duration_s, kind)
    if unit == "log2":
        return round_value(np.log2(fs * duration_s), kind)
    if unit == "pow2":
        return round_value(2 ** (fs * duration_s), kind)
    raise ValueError(unit)


# --------------------------------------------------------------------------- STFT
def tukey_periodic(m, alpha):
    """scipy.signal.get_window(('tukey', alpha), m) (fftbins=True => periodic:
    the symmetric m+1 window with its last point dropped)."""
    if alpha <= 0:
        return np.ones(m)
    mm = m + 1
    n = np.arange(0, mm)
    if alpha >= 1.0:
        # scipy.signal.windows: tukey(alpha>=1) -> hann -> general_cosine(mm, [0.5, 0.5])
        fac = np.linspace(-np.pi, np.pi, mm)
        w = np.zeros(mm)
        for k, a in enumerate([0.5, 0.5]):
            w += a * np.cos(k * fac)
        return w[:-1]
    width = int(np.floor(alpha * (mm - 1) / 2.0))
    n1, n2, n3 = n[0 : width + 1], n[width + 1 : mm - width - 1], n[mm - width - 1 :]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (mm - 1))))
    w2 = np.ones(n2.shape)
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (mm - 1))))
    return np.concatenate((w1, w2, w3))[:-1]


def stft_spectral(sig, fs, window, seg, overlap, nfft):
    """Restatement of scipy.signal.stft(..., detrend='constant', return_onesided=True,
    boundary='zeros', padded=True, scaling='spectrum') (SciPy 1.15.3
    signal/_spectral_py.py:_spectral_helper/_fft_helper) on the last axis.
    Call site: ref styx_fft.py:175-187."""
    x = np.asarray(sig)
    out_dtype = np.result_type(x, np.complex64)
    step = seg - overlap
    half = seg // 2
    x = np.concatenate((np.zeros(x.shape[:-1] + (half,)), x, np.zeros(x.shape[:-1] + (half,))), axis=-1)
    nadd = (-(x.shape[-1] - seg) % step) % seg
    x = np.concatenate((x, np.zeros(x.shape[:-1] + (nadd,))), axis=-1)
    win = np.asarray(window, dtype=np.float64)
    if np.result_type(win, np.complex64) != out_dtype:
        win = win.astype(out_dtype)
    scale = np.sqrt(1.0 / win.sum() ** 2)
    # strided view as in _fft_helper: the per-segment mean is then summed in the same order
    frames = np.lib.stride_tricks.sliding_window_view(x, seg, axis=-1)[..., 0::step, :]
    frames = frames - np.mean(frames, axis=-1, keepdims=True)
    frames = (win * frames).real
    spec = _rfft(frames, n=nfft, axis=-1) * scale
    t = np.arange(seg / 2, x.shape[-1] - seg / 2 + 1, step) / float(fs) - (seg / 2) / fs
    f = np.fft.rfftfreq(nfft, 1 / fs)
    return f, t, np.moveaxis(spec.astype(out_dtype), -1, -2)


def stft_complex_pow2(sig, fs, seg, overlap=None, nfft=None, alpha=0.25):
    """ref: styx_fft.py:152-187."""
    if nfft is None:
        nfft = int(2 ** np.ceil(np.log2(seg)))
    if overlap is None:
        overlap = int(seg / 2)
    return stft_spectral(sig, fs, tukey_periodic(seg, alpha), seg, overlap, nfft)


def gaussian_periodic(m, sigma):
    """scipy.signal.get_window(("gaussian", sigma), m): fftbins=True -> the symmetric (m + 1)-point Gaussian without
    its last sample (SciPy 1.15.3 signal/windows/_windows.py: gaussian, _extend/_truncate)."""
    k = np.arange(0, m + 1) - m / 2.0
    return np.exp(-(k ** 2) / (2 * sigma * sigma))[:-1]


def gtx_complex_pow2(sig, fs, seg, sigma=None, overlap=None, nfft=None):
    """Returns (f, t, Z): the Gaussian-window STFT.  ref: styx_fft.py:190-227."""
    if nfft is None:
        nfft = int(2 ** np.ceil(np.log2(seg)))
    if overlap is None:
        overlap = int(seg / 2)
    if sigma is None:
        sigma = int(seg / 4)
    return stft_spectral(sig, fs, gaussian_periodic(int(seg), sigma), int(seg), overlap, nfft)


def stft_segment_points(fs, order, center_hz=None, octaves_below=4):
    """ref: styx_fft.py:31-41."""
    if center_hz is None:
        center_hz = fs * 0.075
    duration = cycles_from_order(order) / (center_hz / octaves_below)
    return 2 ** get_num_points(fs, duration, "ceil", "log2")


def stft_from_sig(sig, fs, order, center_hz=None, octaves_below=4):
    """Returns (Z, bits, t, f).  ref: styx_fft.py:14-57."""
    seg = stft_segment_points(fs, order, center_hz, octaves_below)
    if len(sig) < seg:
        raise ValueError(f"Signal length: {len(sig)} is less than time_fft_nd: {seg}")
    f, t, z = stft_complex_pow2(sig, fs, seg, alpha=1.0)
    z *= 2 * np.sqrt(np.pi) / seg
    return z, to_log2_with_epsilon(z), t, f


def welch_power_pow2(sig, fs, seg, nfft=None, overlap=None, alpha=0.25):
    """scipy.signal.welch(window=("tukey", alpha), detrend="constant", scaling="spectrum", average="mean",
    return_onesided=True) restated (SciPy 1.15.3 _spectral_helper with boundary=None, padded=False).
    ref: styx_fft.py:230-266."""
    x = np.asarray(sig)
    if nfft is None:
        nfft = int(2 ** np.ceil(np.log2(seg)))
    if overlap is None:
        overlap = int(seg / 2)
    out_dtype = np.result_type(x, np.complex64)
    step = seg - overlap
    win = tukey_periodic(seg, alpha)
    if np.result_type(win, np.complex64) != out_dtype:
        win = win.astype(out_dtype)
    scale = 1.0 / win.sum() ** 2
    frames = np.lib.stride_tricks.sliding_window_view(x, seg, axis=-1)[..., 0::step, :]
    frames = frames - np.mean(frames, axis=-1, keepdims=True)
    frames = (win * frames).real
    spec = _rfft(frames, n=nfft, axis=-1)
    p = np.conjugate(spec) * spec * scale
    if nfft % 2:
        p[..., 1:] *= 2
    else:
        p[..., 1:-1] *= 2
    p = p.astype(out_dtype).real
    return np.fft.rfftfreq(nfft, 1 / fs), p.mean(axis=-2)


# --------------------------------------------------------------------------- styx_cwt
def wavelet_amplitude(scale):
    """ref: styx_cwt.py:29-40."""
    a_norm = (np.pi * scale ** 2) ** (-1 / 4)
    return a_norm, (4 * np.pi * scale ** 2) ** (-1 / 4) * a_norm


def gabor_atom_row(order, n, f_hz, fs, dict_type="norm"):
    """One row of the centred Gabor bank.  ref: styx_cwt.py:68-144, evaluated per
    band (the reference tiles [B x n]; elementwise results are identical)."""
    t = np.arange(n) / fs
    x = fs * (t - t[-1] / 2.0)  # ref: styx_cwt.py:58-65,133-136
    s, w = scale_omega(order, f_hz, fs)
    a_norm, a_spect = wavelet_amplitude(s)
    amp = a_spect if dict_type == "spect" else (1.0 if dict_type == "unit" else a_norm)
    return amp * (np.exp(-0.5 * (x / s) ** 2) * np.exp(1j * w * x))


def cwt_fft(order, sig, fs, dict_type="norm", bands=None):
    """Returns (f, t, cwt).  ref: styx_cwt.py:147-198 (cwt_type='fft').
    scipy.signal.fftconvolve(mode='same', axes=-1) restated: zero-pad both operands
    to L = next_fast_len(2n-1) (= 2n for n a power of two), multiply the spectra,
    inverse, keep [(n-1)//2 : (n-1)//2 + n] (SciPy 1.15.3 signal/_signaltools.py:
    _freq_domain_conv + _centered).  The signal spectrum is computed once instead of
    once per tiled row; SURVEY.md s3.1 measured the two bit-identical for n = 2^k."""
    sig = np.asarray(sig)
    n = len(sig)
    f = band_table(fs, n, order)
    if n & (n - 1) == 0:
        big = 2 * n
    else:
        big = _sfft.next_fast_len(2 * n - 1, False)
    spec = _fft(sig, big)
    rows = range(len(f)) if bands is None else bands
    out = np.empty((len(f) if bands is None else len(bands), n), dtype=np.complex128)
    start = (n - 1) // 2
    for i, j in enumerate(rows):
        h = np.conj(gabor_atom_row(order, n, f[j], fs, dict_type)[::-1])
        out[i] = _ifft(spec * _fft(h, big))[start : start + n]
    return f, np.arange(n) / fs, out


# --------------------------------------------------------------------------- styx_stx
def stx_indices(f_hz, n, fs):
    """argmin_k |fftfreq_k - f_j| (first occurrence).  ref: styx_stx.py:216,233."""
    freq = np.fft.fftfreq(n, 1 / fs)
    return np.array([int(np.abs(freq - fj).argmin()) for fj in f_hz], dtype=np.int64)


def stx_fft(order, sig, fs, bands=None):
    """Returns (f, t, stx).  ref: styx_stx.py:195-236."""
    sig = np.asarray(sig)
    n = len(sig)
    f = band_table(fs, n, order)
    spec = _fft(sig)
    cat = np.concatenate([spec, spec])
    omega_fft = 2 * np.pi * np.fft.fftfreq(n, 1 / fs) / fs
    sigma = cycles_from_order(order) / (2 * np.pi * f / fs)
    idx = stx_indices(f, n, fs)
    rows = range(len(f)) if bands is None else bands
    out = np.empty((len(f) if bands is None else len(bands), n), dtype=np.complex128)
    for i, j in enumerate(rows):
        win = np.exp(-0.5 * (sigma[j] ** 2.0) * (omega_fft ** 2.0))
        out[i] = _ifft(cat[idx[j] : idx[j] + n] * win)
    return f, np.arange(n) / fs, out


def stx_general(sig, dt, order=8.0, f_min=None, f_max=None, f_step=None, q=0.0, p=0.0, r=1.0, geometric=False,
                inferno=False, base=G3, ref_s=1.0):
    """General Stockwell transform for a record of 2^k points with n_fft = len(sig) (the working subset of
    ref styx_stx.py:52-192: its n_fft=None and zero-pad paths raise TypeError upstream).
    Returns (tfr, psd, f_stx, f_stx_fft, windows)."""
    sig = np.asarray(sig)
    n = len(sig)
    fs = 1 / dt
    cycles_m = 12.0 / 5.0 * order
    spec = _fft(sig)
    cat = np.concatenate([spec, spec], axis=-1)
    freq = _sfft.fftfreq(n, dt)
    omega_fft = 2 * np.pi * freq / fs
    if f_min is None:
        f_min = cycles_m / (n / fs)
    if f_max is None:
        f_max = fs / 2.0
    f_start = freq[np.abs(freq - f_min).argmin()]
    f_stop = freq[np.abs(freq - f_max).argmin()]
    if f_step is None:
        f_step = (f_max - f_min) * 2.0 / len(freq)
    f_stx = np.arange(f_start, f_stop, f_step)
    if geometric:
        if inferno:
            f_stx = band_frequency_low_high(order, base, ref_s, f_start, f_stop, fs)[5]
        else:
            f_stx = np.logspace(np.log2(f_start), np.log2(f_stop), num=int(np.log2(f_stop / f_start) * order), base=base)
    f_snap = np.empty(len(f_stx))
    win = np.empty((len(f_stx), n), dtype=np.complex128)
    tfr = np.empty((len(f_stx), n), dtype=np.complex128)
    psd = np.empty((len(f_stx), n))
    for i, f in enumerate(f_stx):
        k = np.abs(freq - f).argmin()
        f_snap[i] = freq[k]
        w = 2 * np.pi * f_snap[i] / fs
        sigma = cycles_m / w * ((1 + q * (w ** p)) * (w ** (1 - r)))
        win[i] = np.exp(-0.5 * (sigma ** 2.0) * (omega_fft ** 2.0))
        tfr[i] = _ifft(cat[k : k + n] * win[i])
        psd[i] = np.abs(tfr[i]) ** 2 + EPS64
    return tfr, psd, f_stx, f_snap, win


# --------------------------------------------------------------------------- cwt_atoms
def chirp_mqg_from_n(order, index_shift=0.0, base=G2):
    """(M_q, Q, gamma).  ref: cwt_atoms.py:122-144."""
    if order < 0.7:
        order = 3.0
    edge = base ** (1.0 / 2.0 / order)
    q = 1.0 / (edge - 1.0 / edge)
    gamma = np.sqrt(np.log(2)) * (1 - np.log(2) * (index_shift / np.pi) ** 2) ** (-0.5)
    return 2 * q * gamma, q, gamma


def chirp_atom(order, n, f_hz, fs, index_shift=0.0, base=G2, dict_type="norm"):
    """Centred chirp/Gabor atom.  ref: cwt_atoms.py:16-50,147-158,202-238,303-340."""
    t = np.arange(n) / fs
    x = fs * (t - t[-1] / 2.0)
    m_q, _, gamma = chirp_mqg_from_n(order, index_shift, base)
    s = m_q * fs / f_hz / (2.0 * np.pi)
    p = (1 - 1j * index_shift * gamma / np.pi) / (2 * s ** 2)
    a_norm = 1 / np.pi ** 0.25 * 1 / np.sqrt(s)
    a_spect = np.sqrt(np.abs(p) / np.pi)
    atom = np.exp(-p * x ** 2) * np.exp(1j * m_q * x / s)
    return (a_norm if dict_type == "norm" else a_spect) * atom


def chirp_band_table(order, n, fs, index_shift=0.0, ref_hz=1.0, base=G2):
    """Descending-period (ascending after flip) band centres used by cwt_chirp_from_sig.
    ref: cwt_atoms.py:241-300,385-399,472-474."""
    m_q, _, _ = chirp_mqg_from_n(order, index_shift, base)
    f_low = 1 / ((n / fs) / m_q)
    f_high = fs / 2.0
    order_n, _, _, _, _, f_geo, _, _ = band_frequency_low_high(order, base, ref_hz, f_low, f_high, fs)
    return order_n, f_geo


def cwt_chirp_fft(sig, fs, order=3, index_shift=0.0, ref_hz=1.0, base=G2, dict_type="norm", bands=None):
    """Returns (cwt, bits, t, f), circular-correlation back-end.
    ref: cwt_atoms.py:343-444 (cwt_type='fft'), :447-486."""
    sig = np.asarray(sig)
    n = len(sig)
    order_n, f_flipped = chirp_band_table(order, n, fs, index_shift, ref_hz, base)
    spec = _np_fft(sig)
    f = np.flip(f_flipped)
    rows = range(len(f)) if bands is None else bands
    out = np.empty((len(f) if bands is None else len(bands), n), dtype=np.complex128)
    for i, j in enumerate(rows):
        atom = chirp_atom(order_n, n, f[j], fs, index_shift, base, dict_type)
        raw = _np_ifft(spec * np.conj(_np_fft(atom)))
        out[i] = np.append(raw[n // 2 :], raw[0 : n // 2])
    return out, to_log2_with_epsilon(out), np.arange(n) / fs, f


# --------------------------------------------------------------------------- tfr_info
def scale_power_bits(power):
    """log2(P + eps64) - max.  ref: tfr_info.py:65-79."""
    bits = np.log2(power + EPS64)
    return bits - np.max(bits)


def power_dynamics_scaled_bits(power):
    """ref: tfr_info.py:82-94."""
    return (
        scale_power_bits(power),
        scale_power_bits(np.sum(power, axis=0)),
        scale_power_bits(np.sum(power, axis=1)),
    )


class ShannonPanel:
    """ref: tfr_info.py:203-228 (class ShannonStft)."""

    def __init__(self, pdf, deg_free):
        self.info = -np.log2(pdf + EPS64)
        self.shannon_bits = pdf * self.info
        self.ref_bits = np.log2(deg_free) / deg_free
        self.isnr = np.log2(deg_free) - self.info
        self.esnr = self.shannon_bits / self.ref_bits


def shannon_from_power(power):
    """ref: tfr_info.py:231-236."""
    return ShannonPanel(power / np.sum(power), power.shape[0] * power.shape[1])


def shannon_per_time(power):
    """pdf = P * (1/sum_axis0 + eps64).  ref: tfr_info.py:239-248, utilities/matrix.py:113-134."""
    return ShannonPanel((1 / np.sum(power, axis=0) + EPS64)[None, :] * power, power.shape[0])


def shannon_per_freq(power):
    """pdf = P * (1/sum_axis1 + eps64).  ref: tfr_info.py:251-260, utilities/matrix.py:89-110."""
    return ShannonPanel((1 / np.sum(power, axis=1) + EPS64)[:, None] * power, power.shape[1])


def shannon_1d(marginal):
    """(info, entropy, ref_entropy, isnr, esnr) with eps32.  ref: tfr_info.py:97-136."""
    info = -np.log2(marginal + EPS32)
    ent = marginal * info
    ref = np.log2(len(marginal)) / len(marginal)
    return info, ent, ref, np.log2(len(info)) - info, ent / ref


def shannon_tdr(sig):
    """(sig_norm, marginal) of ShannonTDR.  ref: tfr_info.py:138-147."""
    sig = np.asarray(sig)
    sig_norm = sig / np.sqrt(np.sum(sig ** 2))
    return sig_norm, sig_norm ** 2


def shannon_fft(sig):
    """(spectrum, angle_rads, frequency, marginal) of ShannonFFT.  ref: tfr_info.py:163-183 (scipy.fft.rfft keeps
    single precision for float32 input)."""
    import scipy.fft as sfft

    spec = sfft.rfft(x=np.asarray(sig))
    angle = np.unwrap(np.angle(spec))
    freq = np.arange(len(angle)) / len(angle) / 2.0
    fft_sq = np.abs(spec) ** 2
    return spec, angle, freq, fft_sq / np.sum(fft_sq)


# --------------------------------------------------------------------------- ShortTimeFFT-convention STFT
def tukey_symmetric(m, alpha):
    """scipy.signal.windows.tukey(m, alpha) (sym=True).  ref call site: utilities/short_time_fft.py:53."""
    if alpha <= 0:
        return np.ones(m)
    n = np.arange(0, m)
    if alpha >= 1.0:
        fac = np.linspace(-np.pi, np.pi, m)
        w = np.zeros(m)
        for k, a in enumerate([0.5, 0.5]):
            w += a * np.cos(k * fac)
        return w
    width = int(np.floor(alpha * (m - 1) / 2.0))
    n1, n2, n3 = n[0 : width + 1], n[width + 1 : m - width - 1], n[m - width - 1 :]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (m - 1))))
    w2 = np.ones(n2.shape)
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (m - 1))))
    return np.concatenate((w1, w2, w3))


class SlidingStft:
    """scipy.signal.ShortTimeFFT (SciPy 1.15.3 signal/_short_time_fft.py) restated for the one configuration the
    reference uses (utilities/short_time_fft.py:20-61): real symmetric Tukey window scaled to 'magnitude' / 'psd',
    fft_mode='onesided', phase_shift=0, mfft = next power of two of the segment."""

    def __init__(self, fs, alpha, seg, overlap, scaling="magnitude"):
        self.T, self.hop, self.m_num, self.m_mid = 1.0 / fs, seg - overlap, seg, seg // 2
        self.mfft = 2 ** int(np.ceil(np.log2(seg)))
        win = tukey_symmetric(seg, alpha)
        if scaling == "magnitude":
            win = win * (1 / abs(sum(win)))
        elif scaling == "psd":
            win = win * (1 / np.sqrt(sum(win ** 2) / self.T))
        self.win = win
        self.f = np.fft.rfftfreq(self.mfft, self.T)
        self.delta_t = self.hop * self.T

    @property
    def p_min(self):
        w2 = self.win ** 2
        n0 = -self.m_mid
        for q_, n_ in enumerate(range(n0, n0 - self.m_num - 1, -self.hop)):
            n_next = n_ - self.hop
            if n_next + self.m_num <= 0 or all(w2[n_next:] == 0):
                return -q_

    def p_max(self, n):
        w2 = self.win ** 2
        q1 = n // self.hop
        k1 = q1 * self.hop - self.m_mid
        for q_, k_ in enumerate(range(k1, n + self.m_num, self.hop), start=q1):
            n_next = k_ + self.hop
            if n_next >= n or all(w2[: n - n_next] == 0):
                return q_ + 1

    def stft(self, x, padding="zeros", detrend=False):
        """S[f, p - p_min] for p_min <= p < p_max(n)  (ShortTimeFFT.stft / stft_detrend(detr='constant'))."""
        import scipy.fft as sfft

        kw = {"zeros": dict(mode="constant", constant_values=(0, 0)), "edge": dict(mode="edge"),
              "even": dict(mode="reflect", reflect_type="even"), "odd": dict(mode="reflect", reflect_type="odd")}[padding]
        n = len(x)
        p0, p1 = self.p_min, self.p_max(n)
        k0 = p0 * self.hop - self.m_mid
        k1 = k0 + (p1 - p0) * self.hop + self.m_num
        x1 = np.pad(x[max(k0, 0) : min(k1, n)], [(-min(k0, 0), max(k1 - n, 0))], **kw)
        p_s = self.m_mid % self.m_num
        out = np.zeros((len(self.f), p1 - p0), dtype=complex)
        for q, k_ in enumerate(range(0, (p1 - p0) * self.hop, self.hop)):
            seg = x1[k_ : k_ + self.m_num]
            if detrend:
                seg = seg - np.mean(seg)
            v = seg * self.win.conj()
            if len(v) < self.mfft:
                v = np.hstack((v, np.zeros(self.mfft - len(v), dtype=v.dtype)))
            out[:, q] = sfft.rfft(np.roll(v, -p_s), n=self.mfft)
        return out

    @property
    def dual_win(self):
        w2 = self.win ** 2
        dd = w2.copy()
        for k_ in range(self.hop, self.m_num, self.hop):
            dd[k_:] += w2[:-k_]
            dd[:-k_] += w2[k_:]
        return self.win / dd

    def istft(self, S, k1):
        """x[0:k1] (ShortTimeFFT.istft(S, k0=0, k1=k1))."""
        import scipy.fft as sfft

        p_s = self.m_mid % self.m_num
        x = np.zeros(k1 + self.m_num)
        for q in range(S.shape[1]):
            xs = np.roll(sfft.irfft(S[:, q], n=self.mfft), p_s)[: self.m_num] * self.dual_win
            i0 = (q + self.p_min) * self.hop - self.m_mid
            lo = max(i0, 0)
            hi = min(i0 + self.m_num, len(x))
            if hi > lo:
                x[lo:hi] += xs[lo - i0 : hi - i0]
        return x[:k1]


def stft_tukey(x, fs, alpha, seg, overlap, scaling="magnitude", padding="zeros"):
    """(f, t, |stft_detrend(x, 'constant')|).  ref: utilities/short_time_fft.py:64-109."""
    o = SlidingStft(fs, alpha, seg, overlap, scaling)
    mag = np.abs(o.stft(x, padding, detrend=True))
    return o.f, np.arange(start=0, stop=o.delta_t * mag.shape[1], step=o.delta_t), mag


def spectrogram_tukey(x, fs, alpha, seg, overlap, scaling="magnitude", padding="zeros"):
    """(f, t, |stft(x)|^2).  ref: utilities/short_time_fft.py:140-175."""
    o = SlidingStft(fs, alpha, seg, overlap, scaling)
    s = o.stft(x, padding, detrend=False)
    sxx = s.real ** 2 + s.imag ** 2
    return o.f, np.arange(start=0, stop=o.delta_t * sxx.shape[1], step=o.delta_t), sxx


def istft_tukey(S, fs, alpha, seg, overlap, scaling="magnitude"):
    """(timestamps, x).  ref: utilities/short_time_fft.py:112-137."""
    o = SlidingStft(fs, alpha, seg, overlap, scaling)
    last = int((S.shape[1] - 1) * o.hop)
    return np.arange(start=0, stop=last / fs, step=1 / fs), o.istft(S, last)


# --------------------------------------------------------------------------- synthetic input
def synth_chirp(n, fs, channel=0, n_channels=1, dtype=np.float32, seed=20250213):
    """Seeded log-chirp test input defined in SURVEY.md s8(d) (this build's own
    generator, not reference code): sin(phase) * tukey(n, 0.05) + noise 8 bits down."""
    k = np.arange(n, dtype=np.float64)
    f0, f1 = fs * 2.0 ** -14, 0.4 * fs
    dur = n / fs
    rate = np.log(f1 / f0) / dur
    phase = 2 * np.pi * f0 * (np.exp(rate * k / fs) - 1.0) / rate + 2 * np.pi * channel / n_channels
    taper = np.ones(n)
    edge = int(np.floor(0.05 * (n - 1) / 2.0))
    ramp = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * np.arange(edge + 1) / 0.05 / (n - 1))))
    taper[: edge + 1] = ramp
    taper[n - edge - 1 :] = ramp[::-1]
    x = np.sin(phase) * taper
    rng = np.random.default_rng(seed + channel)
    x = x + (2.0 ** -8) * np.std(x) * rng.standard_normal(n)
    return x.astype(dtype)
