"""
Short-time Fourier transform on the GPU behind the reference's signatures
(quantum_inferno/styx_fft.py).  The reference calls scipy.signal.stft(boundary="zeros", padded=True,
detrend="constant", return_onesided=True); here the zero-extended segments are mean-removed and
windowed by a HIP kernel, transformed as one batched real FFT and written frequency x time.
"""
from typing import Tuple

import ctypes as C
import numpy as np
import torch

from . import _lib, engine
from .scales_dyadic import cycles_from_order, get_epsilon
from .utilities.calculations import get_num_points


def tukey_window_periodic(points: int, alpha: float) -> np.ndarray:
    """scipy.signal.get_window(("tukey", alpha), points): the periodic (fftbins) Tukey taper, i.e. the
    symmetric window of points + 1 samples without its last one.  alpha >= 1 is the periodic Hann."""
    if alpha <= 0:
        return np.ones(points)
    sym = points + 1
    k = np.arange(0, sym)
    if alpha >= 1.0:
        phase = np.linspace(-np.pi, np.pi, sym)
        return (0.5 + 0.5 * np.cos(phase))[:-1]
    ramp = int(np.floor(alpha * (sym - 1) / 2.0))
    rise = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * k[: ramp + 1] / alpha / (sym - 1))))
    flat = np.ones(sym - 2 * ramp - 2)
    fall = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * k[sym - ramp - 1 :] / alpha / (sym - 1))))
    return np.concatenate((rise, flat, fall))[:-1]


def gaussian_window_periodic(points: int, sigma: float) -> np.ndarray:
    """scipy.signal.get_window(("gaussian", sigma), points) (periodic)."""
    k = np.arange(0, points + 1) - points / 2.0
    return np.exp(-(k ** 2) / (2 * sigma * sigma))[:-1]


def stft_segment_points(frequency_sample_rate_hz, band_order_nth, center_frequency_hz=None, octaves_below_center=4):
    """Power-of-two segment length that holds M cycles of the averaging frequency (ref styx_fft.py:31-41)."""
    if center_frequency_hz is None:
        center_frequency_hz = frequency_sample_rate_hz * 0.075
    duration_s = cycles_from_order(band_order_nth) / (center_frequency_hz / octaves_below_center)
    return 2 ** get_num_points(frequency_sample_rate_hz, duration_s, "ceil", "log2")


def _stft_windowed(sig_wf, fs, window64, segment_points, overlap_points, nfft_points, extra_scale=1.0, want_bits=False):
    lib = _lib.require_gpu()
    sig, was_numpy, was_1d = engine.as_signal(sig_wf)
    n_ch, n = sig.shape
    seg, nfft = int(segment_points), int(nfft_points)
    hop = seg - int(overlap_points)
    if not 0 < hop <= seg:
        raise ValueError("noverlap must be less than nperseg.")
    if nfft < seg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    dev = sig.device
    f64 = sig.dtype == torch.float64
    # SciPy rounds the window to the output precision before using and summing it (_spectral_helper)
    win = window64 if f64 else window64.astype(np.float32)
    scale = float(np.sqrt(1.0 / np.sum(win.astype(np.float64)) ** 2)) * extra_scale
    win_d = torch.from_numpy(np.ascontiguousarray(win)).to(dev)
    n_seg = int(lib.qi_stft_segments(n, seg, hop))
    n_f = nfft // 2 + 1
    cdt = torch.complex128 if f64 else torch.complex64
    z = torch.empty((n_ch, n_f, n_seg), dtype=cdt, device=dev)
    bits = torch.empty((n_ch, n_f, n_seg), dtype=sig.dtype, device=dev) if want_bits else None
    code = _lib.QI_F64 if f64 else _lib.QI_F32
    scratch_bytes = int(lib.qi_stft_scratch_bytes(code, n_ch, n, seg, hop, nfft))
    scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(
            lib.qi_stft(code, dev.index, _lib.ptr(sig), n_ch, n, _lib.ptr(win_d), seg, hop, nfft, scale, _lib.ptr(z),
                        _lib.ptr(bits), float(get_epsilon()), _lib.ptr(scratch), scratch_bytes, _lib.stream_ptr(dev))
        )
    padded = n + 2 * (seg // 2)
    padded += (-(padded - seg) % hop) % seg
    time_s = np.arange(seg / 2, padded - seg / 2 + 1, hop) / float(fs) - (seg / 2) / fs
    freq_hz = np.fft.rfftfreq(nfft, 1 / fs)
    return freq_hz, time_s, engine.finish(z, was_numpy, was_1d), engine.finish(bits, was_numpy, was_1d)


def _shrunk_segment(sig_wf, segment_points):
    """scipy.signal.stft shortens nperseg to the record when the record is shorter, with a warning
    (SciPy 1.15 signal/_spectral_py.py:_triage_segments); the window is then made for the shorter segment."""
    n = sig_wf.shape[-1] if hasattr(sig_wf, "shape") else len(sig_wf)
    if int(segment_points) > n:
        import warnings

        warnings.warn(f"nperseg = {int(segment_points)} is greater than input length  = {n}, using nperseg = {n}")
        return n
    return int(segment_points)


class StftPlan:
    """stft_from_sig for batches of records of one shape with every buffer kept between calls (window on the device,
    outputs, scratch): the per-call work is the kernels alone.  Results are those of stft_from_sig (ref
    styx_fft.py:14-57); `run` returns (stft_complex [C, nfft/2+1, segments], stft_bits)."""

    def __init__(self, n, channels, frequency_sample_rate_hz, band_order_nth, dtype=torch.float32, device=None):
        self._lib = _lib.require_gpu()
        self.n, self.channels, self.fs = int(n), int(channels), float(frequency_sample_rate_hz)
        self.seg = stft_segment_points(self.fs, band_order_nth)
        if self.n < self.seg:
            raise ValueError(f"Signal length: {self.n} is less than time_fft_nd: {self.seg}")
        self.hop, self.nfft = self.seg // 2, self.seg
        self.rdtype = engine._real_dtype(dtype)
        self.device = torch.device(device) if device is not None else engine.default_device()
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        f64 = self.rdtype == torch.float64
        win64 = tukey_window_periodic(self.seg, 1.0)
        win = win64 if f64 else win64.astype(np.float32)
        self.scale = float(np.sqrt(1.0 / np.sum(win.astype(np.float64)) ** 2)) * 2 * np.sqrt(np.pi) / self.seg
        self.window = torch.from_numpy(np.ascontiguousarray(win)).to(self.device)
        self.n_seg = int(self._lib.qi_stft_segments(self.n, self.seg, self.hop))
        self.n_f = self.nfft // 2 + 1
        self.code = _lib.QI_F64 if f64 else _lib.QI_F32
        self.z = torch.empty((self.channels, self.n_f, self.n_seg), dtype=engine._complex_of(self.rdtype), device=self.device)
        self.bits = torch.empty((self.channels, self.n_f, self.n_seg), dtype=self.rdtype, device=self.device)
        self.scratch_bytes = int(self._lib.qi_stft_scratch_bytes(self.code, self.channels, self.n, self.seg, self.hop, self.nfft))
        self.scratch = torch.empty(max(self.scratch_bytes, 1), dtype=torch.uint8, device=self.device)
        padded = self.n + 2 * (self.seg // 2)
        padded += (-(padded - self.seg) % self.hop) % self.seg
        self.time_s = np.arange(self.seg / 2, padded - self.seg / 2 + 1, self.hop) / float(self.fs) - (self.seg / 2) / self.fs
        self.frequency_hz = np.fft.rfftfreq(self.nfft, 1 / self.fs)

    @property
    def points(self):
        """complex coefficients one call produces"""
        return self.channels * self.n_f * self.n_seg

    def run(self, sig):
        if sig.shape != (self.channels, self.n) or sig.dtype != self.rdtype or not sig.is_cuda:
            raise ValueError(f"signal must be a [{self.channels}, {self.n}] {self.rdtype} CUDA tensor")
        with torch.cuda.device(self.device):
            _lib.check(
                self._lib.qi_stft(self.code, self.device.index, _lib.ptr(sig), self.channels, self.n, _lib.ptr(self.window),
                                  self.seg, self.hop, self.nfft, self.scale, _lib.ptr(self.z), _lib.ptr(self.bits),
                                  float(get_epsilon()), _lib.ptr(self.scratch), self.scratch_bytes,
                                  _lib.stream_ptr(self.device))
            )
        return self.z, self.bits


def stft_complex_pow2(
    sig_wf,
    frequency_sample_rate_hz: float,
    segment_points: int,
    overlap_points: int = None,
    nfft_points: int = None,
    alpha: float = 0.25,
):
    """Tukey-window STFT with 50 % overlap by default, last axis (ref styx_fft.py:152-187).
    :return: frequency_stft_hz, time_stft_s, stft_complex [(nfft/2+1) x segments]"""
    if nfft_points is None:
        nfft_points = int(2 ** np.ceil(np.log2(segment_points)))
    if overlap_points is None:
        overlap_points = int(segment_points / 2)
    segment_points = _shrunk_segment(sig_wf, segment_points)
    window = tukey_window_periodic(int(segment_points), alpha)
    f, t, z, _ = _stft_windowed(sig_wf, frequency_sample_rate_hz, window, segment_points, overlap_points, nfft_points)
    return f, t, z


def gtx_complex_pow2(
    sig_wf,
    frequency_sample_rate_hz: float,
    segment_points: int,
    gaussian_sigma: int = None,
    overlap_points: int = None,
    nfft_points: int = None,
):
    """Gaussian-window STFT (ref styx_fft.py:190-227).  :return: frequency_hz, time_s, stft_complex"""
    if nfft_points is None:
        nfft_points = int(2 ** np.ceil(np.log2(segment_points)))
    if overlap_points is None:
        overlap_points = int(segment_points / 2)
    if gaussian_sigma is None:
        gaussian_sigma = int(segment_points / 4)
    segment_points = _shrunk_segment(sig_wf, segment_points)
    window = gaussian_window_periodic(int(segment_points), gaussian_sigma)
    f, t, z, _ = _stft_windowed(sig_wf, frequency_sample_rate_hz, window, segment_points, overlap_points, nfft_points)
    return f, t, z


def stft_from_sig(
    sig_wf,
    frequency_sample_rate_hz: float,
    band_order_nth: float,
    center_frequency_hz: float = None,
    octaves_below_center: int = 4,
):
    """Order-N standardised STFT: periodic-Hann segments sized from N, scaled by 2 sqrt(pi) / segment,
    with its log2 amplitude bits (ref styx_fft.py:14-57).
    :return: stft_complex, stft_bits, time_stft_s, frequency_stft_hz  (note the order)"""
    seg = stft_segment_points(frequency_sample_rate_hz, band_order_nth, center_frequency_hz, octaves_below_center)
    n = sig_wf.shape[-1] if hasattr(sig_wf, "shape") else len(sig_wf)
    if n < seg:
        raise ValueError(f"Signal length: {n} is less than time_fft_nd: {seg}")
    window = tukey_window_periodic(seg, 1.0)
    f, t, z, bits = _stft_windowed(
        sig_wf, frequency_sample_rate_hz, window, seg, seg // 2, seg, extra_scale=2 * np.sqrt(np.pi) / seg,
        want_bits=True,
    )
    return z, bits, t, f


def welch_power_pow2(
    sig_wf,
    frequency_sample_rate_hz: float,
    segment_points: int,
    nfft_points: int = None,
    overlap_points: int = None,
    alpha: float = 0.25,
):
    """Welch power spectrum: mean over 50 %-overlapped Tukey segments of the one-sided |FFT|^2, "spectrum" scaling
    (ref styx_fft.py:230-266).  :return: frequency_welch_hz, welch_power [nfft/2+1] (or [channels x ...])"""
    lib = _lib.require_gpu()
    if nfft_points is None:
        nfft_points = int(2 ** np.ceil(np.log2(segment_points)))
    if overlap_points is None:
        overlap_points = int(segment_points / 2)
    sig, was_numpy, was_1d = engine.as_signal(sig_wf)
    n_ch, n = sig.shape
    seg, nfft = int(segment_points), int(nfft_points)
    hop = seg - int(overlap_points)
    if n < seg:
        raise ValueError(f"Signal length: {n} is less than the segment: {seg}")
    if not 0 < hop <= seg:
        raise ValueError("noverlap must be less than nperseg.")
    f64 = sig.dtype == torch.float64
    win64 = tukey_window_periodic(seg, alpha)
    win = win64 if f64 else win64.astype(np.float32)
    scale = float(1.0 / np.sum(win.astype(np.float64)))
    win_d = torch.from_numpy(np.ascontiguousarray(win)).to(sig.device)
    code = _lib.QI_F64 if f64 else _lib.QI_F32
    pxx = torch.empty((n_ch, nfft // 2 + 1), dtype=sig.dtype, device=sig.device)
    nbytes = int(lib.qi_welch_scratch_bytes(code, n_ch, n, seg, hop, nfft))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=sig.device)
    with torch.cuda.device(sig.device):
        _lib.check(lib.qi_welch(code, sig.device.index, _lib.ptr(sig), n_ch, n, _lib.ptr(win_d), seg, hop, nfft, scale,
                                _lib.ptr(pxx), _lib.ptr(scratch), nbytes, _lib.stream_ptr(sig.device)))
    return np.fft.rfftfreq(nfft, 1 / frequency_sample_rate_hz), engine.finish(pxx, was_numpy, was_1d)
