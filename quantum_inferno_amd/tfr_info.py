"""
Power, information and entropy of a TFR power panel on the GPU, behind the reference's
signatures (quantum_inferno/tfr_info.py).  Every function is a reduction (max / row sums /
column sums / total, one pass, fixed summation order) followed by an elementwise map; the
np.tile broadcasts of the reference become in-register broadcasts.

Panels are [bands x time] as in the reference; a leading channel axis is accepted too.
NumPy in -> NumPy out, CUDA tensor in -> CUDA tensors out, float32 stays float32.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, engine
from . import scales_dyadic as scales

_EPS = float(scales.EPSILON64)


def _as_panel(x):
    """-> (tensor [C, B, n] on the GPU, was_numpy, original ndim)."""
    was_numpy = not isinstance(x, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x))) if was_numpy else x
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if not t.is_cuda:
        t = t.to(engine.default_device())
    nd = t.dim()
    if nd == 1:
        t = t.reshape(1, 1, -1)
    elif nd == 2:
        t = t.unsqueeze(0)
    elif nd != 3:
        raise TypeError(f"Cannot handle an array of shape {tuple(t.shape)}.")
    return t.contiguous(), was_numpy, nd


def _back(t, was_numpy, nd):
    if t is None:
        return None
    if nd == 1:
        t = t.reshape(-1)
    elif nd == 2:
        t = t[0]
    return t.cpu().numpy() if was_numpy else t


def _code(t):
    return _lib.QI_F64 if t.dtype == torch.float64 else _lib.QI_F32


def power_marginals(power):
    """One pass over P [C, B, n]: (sum over time [C, B] f64, sum over bands [C, n], stats [C, 4] f64 =
    max, total, sum P log2 P, 0).  The sums / max that tfr_info.py:82-94,231 take before the log2."""
    lib = _lib.require_gpu()
    n_ch, n_b, n = power.shape
    dev = power.device
    band = torch.empty((n_ch, n_b), dtype=torch.float64, device=dev)
    time = torch.empty((n_ch, n), dtype=power.dtype, device=dev)
    stats = torch.empty((n_ch, 4), dtype=torch.float64, device=dev)
    nbytes = int(lib.qi_power_marginals_scratch_bytes(n_ch, n_b, n))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(
            lib.qi_power_marginals(_code(power), dev.index, _lib.ptr(power), n_ch, n_b, n, _lib.ptr(band),
                                   _lib.ptr(time), _lib.ptr(stats), _lib.ptr(scratch), nbytes, _lib.stream_ptr(dev))
        )
    return band, time, stats


def _log2_offset(x, ref=None, eps=_EPS):
    """log2(x + eps) - ref[c] on [C, count]."""
    lib = _lib.require_gpu()
    out = torch.empty_like(x)
    n_ch = x.shape[0]
    count = x.numel() // n_ch
    with torch.cuda.device(x.device):
        _lib.check(
            lib.qi_log2_offset(_code(x), x.device.index, _lib.ptr(x), _lib.ptr(out), n_ch, count, eps, _lib.ptr(ref),
                               _lib.stream_ptr(x.device))
        )
    return out


def log2_abs(x, eps=_EPS):
    """log2(|x| + eps) of a real or complex CUDA tensor through the library (qi_log2_abs); float32 / complex64 stay
    single precision, everything else is computed in double."""
    lib = _lib.require_gpu()
    cplx = x.is_complex()
    if x.dtype not in (torch.float32, torch.float64, torch.complex64, torch.complex128):
        x = x.to(torch.float64)
    x = x.contiguous()
    rdt = torch.float32 if x.dtype in (torch.float32, torch.complex64) else torch.float64
    out = torch.empty(x.shape, dtype=rdt, device=x.device)
    if x.numel() == 0:
        return out
    src = torch.view_as_real(x) if cplx else x
    with torch.cuda.device(x.device):
        _lib.check(lib.qi_log2_abs(_lib.QI_F64 if rdt == torch.float64 else _lib.QI_F32, x.device.index, _lib.ptr(src),
                                   1 if cplx else 0, _lib.ptr(out), x.numel(), float(eps), _lib.stream_ptr(x.device)))
    return out


def scale_log2_64(in_array):
    """log2(x + eps64) (ref tfr_info.py:65-70)."""
    p, was_numpy, nd = _as_panel(in_array)
    return _back(_log2_offset(p), was_numpy, nd)


def _scaled_bits(p):
    """log2(P + eps) - max(log2(P + eps)) per channel; log2 is monotonic so the max is taken on P."""
    _, _, stats = power_marginals(p)
    return _log2_offset(p, _log2_offset(stats[:, 0:1].contiguous())[:, 0].contiguous())


def scale_power_bits(power):
    """Power bits relative to the panel maximum (ref tfr_info.py:73-79)."""
    p, was_numpy, nd = _as_panel(power)
    return _back(_scaled_bits(p), was_numpy, nd)


def power_dynamics_scaled_bits(tfr_power):
    """(power bits re max [B x n], per-time bits [n], per-frequency bits [B]) (ref tfr_info.py:82-94)."""
    p, was_numpy, nd = _as_panel(tfr_power)
    band, time, stats = power_marginals(p)
    bits = _log2_offset(p, _log2_offset(stats[:, 0:1].contiguous())[:, 0].contiguous())
    per_time = _scaled_bits(time.unsqueeze(1))[:, 0]
    per_freq = _scaled_bits(band.to(p.dtype).unsqueeze(1))[:, 0]
    squeeze = (lambda t: t[0]) if nd <= 2 else (lambda t: t)
    conv = (lambda t: t.cpu().numpy()) if was_numpy else (lambda t: t)
    return _back(bits, was_numpy, nd), conv(squeeze(per_time)), conv(squeeze(per_freq))


class ShannonStft:
    """Shannon information of a TFR probability panel (ref tfr_info.py:203-228):
    info = -log2(pdf + eps64), shannon_bits = pdf * info, ref_bits = log2(D) / D,
    isnr = log2(D) - info, esnr = shannon_bits / ref_bits."""

    def __init__(self, tfr_pow_pdf, deg_free: int, _mult=None, _mode=0):
        lib = _lib.require_gpu()
        p, was_numpy, nd = _as_panel(tfr_pow_pdf)
        n_ch, n_b, n = p.shape
        if _mult is None:
            _mult = torch.ones(n_ch, dtype=p.dtype, device=p.device)
        mult = _mult.to(p.dtype).contiguous()
        outs = [torch.empty_like(p) for _ in range(4)]
        with torch.cuda.device(p.device):
            _lib.check(
                lib.qi_shannon_panel(_code(p), p.device.index, _lib.ptr(p), _lib.ptr(mult), _mode, n_ch, n_b, n,
                                     float(deg_free), *[_lib.ptr(o) for o in outs], _lib.stream_ptr(p.device))
            )
        self.info, self.shannon_bits, self.isnr, self.esnr = (_back(o, was_numpy, nd) for o in outs)
        self.ref_bits: float = np.log2(deg_free) / deg_free


def shannon_stft_from_tfr_power(tfr_power) -> ShannonStft:
    """ShannonStft of P / sum(P) with D = bands * times (ref tfr_info.py:231-236)."""
    p, _, _ = _as_panel(tfr_power)
    _, _, stats = power_marginals(p)
    return ShannonStft(tfr_power, p.shape[1] * p.shape[2], _mult=1.0 / stats[:, 1], _mode=0)


class ShannonStftPerTime(ShannonStft):
    """pdf = P * (1 / sum over bands + eps64), D = bands (ref tfr_info.py:239-248)."""

    def __init__(self, tfr_power):
        p, _, _ = _as_panel(tfr_power)
        _, time, _ = power_marginals(p)
        super().__init__(tfr_power, p.shape[1], _mult=1.0 / time.to(torch.float64) + _EPS, _mode=1)


class ShannonStftPerFreq(ShannonStft):
    """pdf = P * (1 / sum over time + eps64), D = times (ref tfr_info.py:251-260)."""

    def __init__(self, tfr_power):
        p, _, _ = _as_panel(tfr_power)
        band, _, _ = power_marginals(p)
        super().__init__(tfr_power, p.shape[2], _mult=1.0 / band + _EPS, _mode=2)


# ---- 1-D Shannon information of a record and of its spectrum (tfr_info.py:97-200) -----------------------------------
def _as_rows(x):
    """-> (tensor [C, n] on the GPU, was_numpy, was_1d)."""
    was_numpy = not isinstance(x, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x))) if was_numpy else x
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if not t.is_cuda:
        t = t.to(engine.default_device())
    one = t.dim() == 1
    if one:
        t = t.unsqueeze(0)
    if t.dim() != 2:
        raise TypeError(f"Cannot handle an array of shape {tuple(t.shape)}.")
    return t.contiguous(), was_numpy, one


def _rows_back(t, was_numpy, one):
    if one:
        t = t[0]
    return t.cpu().numpy() if was_numpy else t


def _shannon_1d(marginal):
    """(info, entropy, isnr, esnr) of marginals [C, n] (qi_shannon_1d)."""
    lib = _lib.require_gpu()
    n_ch, n = marginal.shape
    outs = [torch.empty_like(marginal) for _ in range(4)]
    with torch.cuda.device(marginal.device):
        _lib.check(lib.qi_shannon_1d(_code(marginal), marginal.device.index, _lib.ptr(marginal), n_ch, n,
                                     *[_lib.ptr(o) for o in outs], _lib.stream_ptr(marginal.device)))
    return outs


def get_info_and_entropy_32(marginal):
    """info, entropy and reference entropy of a 1-D marginal with EPSILON32 (tfr_info.py:97-104)."""
    m, was_numpy, one = _as_rows(marginal)
    info, entropy, _, _ = _shannon_1d(m)
    n = m.shape[1]
    return _rows_back(info, was_numpy, one), _rows_back(entropy, was_numpy, one), np.log2(n) / n


class Shannon:
    """Shannon information of a 1-D marginal (tfr_info.py:107-133): marginal, info, entropy, ref_entropy, isnr, esnr.
    A leading channel axis is accepted."""

    def __init__(self, marginal, _rows=None):
        m, was_numpy, one = _rows if _rows is not None else _as_rows(marginal)
        info, entropy, isnr, esnr = _shannon_1d(m)
        back = lambda t: _rows_back(t, was_numpy, one)  # noqa: E731
        self.marginal = back(m)
        self.info = back(info)
        self.entropy = back(entropy)
        self.ref_entropy = np.log2(m.shape[1]) / m.shape[1]
        self.isnr = back(isnr)
        self.esnr = back(esnr)


def _scratch(lib, t):
    nbytes = int(lib.qi_shannon_scratch_bytes(_code(t), t.shape[0], t.shape[1]))
    return torch.empty(nbytes, dtype=torch.uint8, device=t.device), nbytes


class ShannonTDR(Shannon):
    """Shannon information of the normalised record (tfr_info.py:136-158): sig = x / sqrt(sum x^2), marginal = sig^2."""

    def __init__(self, sig_in_real):
        lib = _lib.require_gpu()
        x, was_numpy, one = _as_rows(sig_in_real)
        sig = torch.empty_like(x)
        marginal = torch.empty_like(x)
        scratch, nbytes = _scratch(lib, x)
        with torch.cuda.device(x.device):
            _lib.check(lib.qi_shannon_tdr(_code(x), x.device.index, _lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(sig),
                                          _lib.ptr(marginal), _lib.ptr(scratch), nbytes, _lib.stream_ptr(x.device)))
        self.sig = _rows_back(sig, was_numpy, one)
        super().__init__(None, _rows=(marginal, was_numpy, one))

    def print_total_ref_entropy(self):
        print("Ref entropy, time:", self.ref_entropy)

    def print_total_entropy(self):
        print("Total Entropy, time:", self.entropy.sum())

    def print_total_marginal(self):
        print("Sum of time marginal:", self.marginal.sum())


class ShannonFFT(Shannon):
    """Shannon information of the spectrum (tfr_info.py:161-188): sig = rfft(x), angle_rads = unwrap(angle(sig)),
    frequency = arange / len / 2, marginal = |sig|^2 / sum |sig|^2."""

    def __init__(self, sig_in_real):
        lib = _lib.require_gpu()
        x, was_numpy, one = _as_rows(sig_in_real)
        n_ch, n = x.shape
        nf = n // 2 + 1
        spec = torch.empty((n_ch, nf), dtype=engine._complex_of(x.dtype), device=x.device)
        angle = torch.empty((n_ch, nf), dtype=x.dtype, device=x.device)
        marginal = torch.empty((n_ch, nf), dtype=x.dtype, device=x.device)
        scratch, nbytes = _scratch(lib, x)
        with torch.cuda.device(x.device):
            _lib.check(lib.qi_shannon_fft(_code(x), x.device.index, _lib.ptr(x), n_ch, n, _lib.ptr(spec), _lib.ptr(angle),
                                          _lib.ptr(marginal), _lib.ptr(scratch), nbytes, _lib.stream_ptr(x.device)))
        self.sig = _rows_back(spec, was_numpy, one)
        self.angle_rads = _rows_back(angle, was_numpy, one)
        self.frequency = np.arange(nf) / nf / 2.0
        super().__init__(None, _rows=(marginal, was_numpy, one))

    def print_total_ref_entropy(self):
        print("Ref entropy, frequency:", self.ref_entropy)

    def print_total_entropy(self):
        print("Total Entropy, frequency:", self.entropy.sum())

    def print_total_marginal(self):
        print("Sum of frequency marginal:", self.marginal.sum())


def shannon_tdr_fft(sig_in_real):
    """ShannonTDR and ShannonFFT of the record (tfr_info.py:191-200)."""
    return ShannonTDR(sig_in_real), ShannonFFT(sig_in_real)
