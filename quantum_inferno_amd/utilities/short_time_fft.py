"""
Tukey-window STFT / spectrogram / inverse STFT in scipy.signal.ShortTimeFFT's convention, on the GPU, behind the
signatures of quantum_inferno/utilities/short_time_fft.py:20-175.

The reference builds a scipy.signal.ShortTimeFFT object (symmetric Tukey window, mfft = next power of two of the
segment, one-sided, window scaled to "magnitude" or "psd") and calls its stft_detrend / spectrogram / istft.  Here the
slice geometry of that class (p_min, p_max, k_max, the canonical dual window; SciPy 1.15 signal/_short_time_fft.py) is
restated on the host in a small descriptor, `TukeyStft`, and the transforms run through qi_sliding_stft /
qi_sliding_istft.  NumPy in -> NumPy out, CUDA tensor in -> CUDA tensor out; a leading channel axis is accepted.
"""
import numpy as np
import torch

from .. import _lib, engine
from .calculations import round_value

scaling_type = ["magnitude", "psd", None]
padding_type = ["zeros", "edge", "even", "odd"]
_PAD_CODE = {"zeros": 0, "edge": 1, "even": 2, "odd": 3}


def tukey_window_symmetric(points: int, alpha: float) -> np.ndarray:
    """scipy.signal.windows.tukey(points, alpha) (sym=True): cosine tapers of alpha/2 of the length at both ends."""
    if points < 1:
        return np.array([])
    if points == 1:
        return np.ones(1)
    if alpha <= 0:
        return np.ones(points)
    if alpha >= 1.0:
        fac = np.linspace(-np.pi, np.pi, points)  # scipy: tukey(alpha >= 1) -> hann -> general_cosine([0.5, 0.5])
        return 0.5 * np.cos(0 * fac) + 0.5 * np.cos(fac)
    n = np.arange(0, points)
    width = int(np.floor(alpha * (points - 1) / 2.0))
    n1, n2, n3 = n[0 : width + 1], n[width + 1 : points - width - 1], n[points - width - 1 :]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (points - 1))))
    w2 = np.ones(n2.shape)
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (points - 1))))
    return np.concatenate((w1, w2, w3))


class TukeyStft:
    """What the reference's get_stft_object_tukey returns, reduced to the numbers the transforms need: the scaled
    window `win`, `hop`, `mfft`, `fs`, the frequencies `f`, `delta_t`, and ShortTimeFFT's slice geometry."""

    def __init__(self, sample_rate_hz, window, hop, mfft, scaling):
        self.fs = float(sample_rate_hz)
        self.T = 1.0 / self.fs
        self.hop = int(hop)
        self.mfft = int(mfft)
        self.m_num = len(window)
        self.m_num_mid = self.m_num // 2
        self.scaling = scaling
        win = np.asarray(window, dtype=np.float64)
        # ShortTimeFFT.scale_to: the window itself carries the factor (SciPy sums with the builtin, in index order)
        if scaling == "magnitude":
            win = win * (1 / abs(sum(win)))
        elif scaling == "psd":
            win = win * (1 / np.sqrt(sum(win.real ** 2 + win.imag ** 2) / self.T))
        self.win = win
        self.delta_t = self.hop * self.T
        self.f = np.fft.rfftfreq(self.mfft, self.T)
        self.f_pts = len(self.f)
        self._dual = None

    # -- slice geometry (ShortTimeFFT._pre_padding / _post_padding) ------------------------------------------------
    @property
    def p_min(self):
        w2 = self.win ** 2
        n0 = -self.m_num_mid
        for q_, n_ in enumerate(range(n0, n0 - self.m_num - 1, -self.hop)):
            n_next = n_ - self.hop
            if n_next + self.m_num <= 0 or np.all(w2[n_next:] == 0):
                return -q_
        raise RuntimeError("window never leaves the record start")

    def _post_padding(self, n):
        if not n >= self.m_num - self.m_num_mid:
            raise ValueError(f"Parameter n must be >= ceil(m_num/2) = {self.m_num - self.m_num_mid}!")
        w2 = self.win ** 2
        q1 = n // self.hop
        k1 = q1 * self.hop - self.m_num_mid
        for q_, k_ in enumerate(range(k1, n + self.m_num, self.hop), start=q1):
            n_next = k_ + self.hop
            if n_next >= n or np.all(w2[: n - n_next] == 0):
                return k_ + self.m_num, q_ + 1
        raise RuntimeError("window never leaves the record end")

    def p_max(self, n):
        return self._post_padding(n)[1]

    def k_max(self, n):
        return self._post_padding(n)[0]

    @property
    def k_min(self):
        return self.p_min * self.hop - self.m_num_mid

    @property
    def dual_win(self):
        """Canonical dual window win / sum_k |win shifted by k hop|^2 (ShortTimeFFT.dual_win)."""
        if self._dual is None:
            if self.hop > self.m_num:
                raise ValueError(f"hop={self.hop} is larger than window length of {self.m_num} => STFT not invertible!")
            w2 = self.win ** 2
            dd = w2.copy()
            for k_ in range(self.hop, self.m_num, self.hop):
                dd[k_:] += w2[:-k_]
                dd[:-k_] += w2[k_:]
            if not np.all(dd >= np.finfo(np.float64).resolution * max(dd)):
                raise ValueError("Short-time Fourier Transform not invertible!")
            self._dual = self.win / dd
        return self._dual


def get_stft_object_tukey(sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling="magnitude") -> TukeyStft:
    """short_time_fft.py:20-61: symmetric Tukey window, mfft = ceil_power_of_two(segment), hop = segment - overlap."""
    if segment_length < overlap_length:
        print(f"overlap length {overlap_length} must be smaller than segment length {segment_length}"
              " using half of the segment length as the overlap length")
        overlap_length = segment_length // 2
    if tukey_alpha < 0 or tukey_alpha > 1:
        print(f"Warning: Tukey alpha {tukey_alpha} must be between 0 and 1, using 0.25 as the default value")
        tukey_alpha = 0.25
    if scaling not in scaling_type:
        print(f"Warning: scaling {scaling} must be one of {scaling_type}, using 'magnitude' as the default value")
        scaling = "magnitude"
    window = tukey_window_symmetric(segment_length, tukey_alpha)
    return TukeyStft(sample_rate_hz, window, segment_length - overlap_length,
                     round_value(segment_length, "ceil_power_of_two"), scaling)


def _as_rows(x, complex_ok=False):
    was_numpy = not isinstance(x, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x))) if was_numpy else x
    if complex_ok:
        if t.dtype not in (torch.complex64, torch.complex128):
            t = t.to(torch.complex128)
    elif t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if not t.is_cuda:
        t = t.to(engine.default_device())
    return t.contiguous(), was_numpy


def _code(real_dtype):
    return _lib.QI_F64 if real_dtype == torch.float64 else _lib.QI_F32


def _forward(obj: TukeyStft, timeseries, padding, detrend, real_kind):
    """|STFT| (real_kind 1) or |STFT|^2 (2) of the record(s): [.., f_pts, p_num]."""
    lib = _lib.require_gpu()
    x, was_numpy = _as_rows(timeseries)
    one = x.dim() == 1
    if one:
        x = x.unsqueeze(0)
    n_ch, n = x.shape
    p0, p1 = obj.p_min, obj.p_max(n)
    n_slices = p1 - p0
    first = p0 * obj.hop - obj.m_num_mid
    win = torch.from_numpy(obj.win).to(device=x.device, dtype=x.dtype)
    out = torch.empty((n_ch, obj.f_pts, n_slices), dtype=x.dtype, device=x.device)
    nbytes = int(lib.qi_sliding_scratch_bytes(_code(x.dtype), n_ch, obj.mfft, n_slices))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.qi_sliding_stft(_code(x.dtype), x.device.index, _lib.ptr(x), n_ch, n, _lib.ptr(win), obj.m_num,
                                       obj.hop, obj.mfft, first, n_slices, _PAD_CODE[padding], 1 if detrend else 0,
                                       obj.m_num_mid, None, _lib.ptr(out), real_kind, _lib.ptr(scratch), nbytes,
                                       _lib.stream_ptr(x.device)))
    if one:
        out = out[0]
    return out.cpu().numpy() if was_numpy else out


def stft_tukey(timeseries, sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling="magnitude",
               padding="zeros"):
    """frequency bins, time bins and |STFT| with every slice detrended (mean removed) first: short_time_fft.py:64-109."""
    if padding not in padding_type:
        print(f"Warning: padding {padding} must be one of {padding_type}, using 'zeros' as the default value")
        padding = "zeros"
    obj = get_stft_object_tukey(sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling)
    magnitude = _forward(obj, timeseries, padding, True, 1)
    time_bins = np.arange(start=0, stop=obj.delta_t * magnitude.shape[-1], step=obj.delta_t)
    return obj.f, time_bins, magnitude


def spectrogram_tukey(timeseries, sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling="magnitude",
                      padding="zeros"):
    """frequency bins, time bins and |STFT|^2 (no detrending): short_time_fft.py:140-175."""
    if padding not in padding_type:
        print(f"Warning: padding {padding} must be one of {padding_type}, using 'zeros' as the default value")
        padding = "zeros"
    obj = get_stft_object_tukey(sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling)
    spectrogram = _forward(obj, timeseries, padding, False, 2)
    time_bins = np.arange(start=0, stop=obj.delta_t * spectrogram.shape[-1], step=obj.delta_t)
    return obj.f, time_bins, spectrogram


def istft_tukey(stft_to_invert, sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling="magnitude"):
    """timestamps and the inverse STFT up to the last window index: short_time_fft.py:112-137 (ShortTimeFFT.istft with
    k1 = (columns - 1) * hop).  The STFT must carry ShortTimeFFT's phase convention (phase_shift = 0) and start at
    slice p_min, as ShortTimeFFT.stft returns it."""
    lib = _lib.require_gpu()
    obj = get_stft_object_tukey(sample_rate_hz, tukey_alpha, segment_length, overlap_length, scaling)
    s, was_numpy = _as_rows(stft_to_invert, complex_ok=True)
    one = s.dim() == 2
    if one:
        s = s.unsqueeze(0)
    n_ch, f_pts, n_slices = s.shape
    if f_pts != obj.f_pts:
        raise ValueError(f"S.shape[f_axis]={f_pts} must be equal to self.f_pts={obj.f_pts} (S.shape={tuple(s.shape)})!")
    k0, k1 = 0, int((n_slices - 1) * obj.hop)
    q_max = n_slices + obj.p_min
    k_max = (q_max - 1) * obj.hop + obj.m_num - obj.m_num_mid
    if not (obj.k_min <= k0 < k1 <= k_max):
        raise ValueError(f"(self.k_min={obj.k_min}) <= (k0={k0}) < (k1={k1}) <= (k_max={k_max}) is false!")
    if not (k1 - k0) >= obj.m_num - obj.m_num_mid:
        raise ValueError(f"(k1={k1}) - (k0={k0}) = {k1 - k0} has to be at least the half the window length "
                         f"{obj.m_num - obj.m_num_mid}!")
    rdtype = torch.float64 if s.dtype == torch.complex128 else torch.float32
    dual = torch.from_numpy(obj.dual_win).to(device=s.device, dtype=rdtype)
    out = torch.empty((n_ch, k1 - k0), dtype=rdtype, device=s.device)
    first = obj.p_min * obj.hop - obj.m_num_mid
    nbytes = int(lib.qi_sliding_scratch_bytes(_code(rdtype), n_ch, obj.mfft, n_slices))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        _lib.check(lib.qi_sliding_istft(_code(rdtype), s.device.index, _lib.ptr(s), n_ch, _lib.ptr(dual), obj.m_num, obj.hop,
                                        obj.mfft, first, n_slices, obj.m_num_mid, k0, k1, _lib.ptr(out), _lib.ptr(scratch),
                                        nbytes, _lib.stream_ptr(s.device)))
    if one:
        out = out[0]
    timestamps = np.arange(start=0, stop=k1 / sample_rate_hz, step=1 / sample_rate_hz)
    return timestamps, (out.cpu().numpy() if was_numpy else out)
