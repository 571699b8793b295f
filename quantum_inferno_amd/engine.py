"""
Batched GPU engine behind the reference-signature wrappers.

A `TfrPlan` owns one qi_plan (include/qi_tfr.h): record length n, arithmetic type, device, the
Gabor atom banks and the Stockwell band table.  Band selection is done here on the host in
float64 (scales_dyadic) and handed to the library as tables; everything that touches a panel
runs in libqi_tfr.so.  Signals are [channels, n]; panels are [channels, bands, n].

PyTorch is used for device memory and streams only.
"""
import collections
import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import scales_dyadic as scales

_EPI_SPAN = 1024  # time samples per epilogue workgroup (csrc/qi_common.hpp:kEpiSpan)


def _real_dtype(dtype):
    if dtype in (torch.float32, np.float32, "float32", "f32"):
        return torch.float32
    if dtype in (torch.float64, np.float64, "float64", "f64", float):
        return torch.float64
    raise TypeError(f"unsupported dtype {dtype}: float32 or float64")


def _complex_of(rdtype):
    return torch.complex64 if rdtype == torch.float32 else torch.complex128


def default_device():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def as_signal(sig, device=None, dtype=None):
    """-> (tensor [C, n] on the GPU, was_numpy, was_1d).  float32 stays float32, everything else
    becomes float64 (the reference computes in float64)."""
    was_numpy = not isinstance(sig, torch.Tensor)
    if was_numpy:
        arr = np.asarray(sig)
        if arr.dtype != np.float32 and arr.dtype != np.float64:
            arr = arr.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(arr))
    else:
        t = sig
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float64)
    if dtype is not None:
        t = t.to(_real_dtype(dtype))
    if device is None:
        device = t.device if t.is_cuda else default_device()
    t = t.to(device)
    was_1d = t.dim() == 1
    if was_1d:
        t = t.unsqueeze(0)
    if t.dim() != 2:
        raise ValueError(f"signal must be 1-D [n] or 2-D [channels, n], got shape {tuple(t.shape)}")
    return t.contiguous(), was_numpy, was_1d


def _log2_rows(x, eps=0.0):
    """log2(x + eps) of a [C, count] real device tensor through the library (qi_log2_offset) -- the small marginals too:
    PyTorch computes nothing on the path."""
    lib = _lib.require_gpu()
    x = x.contiguous()
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        _lib.check(lib.qi_log2_offset(_lib.QI_F64 if x.dtype == torch.float64 else _lib.QI_F32, x.device.index, _lib.ptr(x),
                                      _lib.ptr(out), x.shape[0], x.numel() // x.shape[0], float(eps), None,
                                      _lib.stream_ptr(x.device)))
    return out


@dataclass
class TfrResult:
    """Outputs of one transform call; every field is a device tensor or None."""

    frequency_hz: np.ndarray
    coef: Optional[torch.Tensor] = None  # [C, B, n] complex
    bits: Optional[torch.Tensor] = None  # [C, B, n] log2(|z| + eps)
    power_band: Optional[torch.Tensor] = None  # [C, B] float64, sum over time of P
    power_time: Optional[torch.Tensor] = None  # [C, n], sum over bands of P
    stats: Optional[torch.Tensor] = None  # [C, 4] float64: max P, sum P, sum P log2 P, 0
    power_scale: float = 1.0
    reduced: Optional[torch.Tensor] = None  # float64 buffer the three reductions are views of (dist.reduced_slots)

    @property
    def max_power(self):
        return self.stats[:, 0]

    @property
    def total_power(self):
        return self.stats[:, 1]

    @property
    def entropy_bits(self):
        """Total Shannon entropy sum(pdf * -log2(pdf)), pdf = P / sum(P), from the one-pass sums
        H = log2 S - (sum P log2 P) / S (tfr_info.py:203-236 without materialising the panel; the
        reference's eps64 inside the log changes H by < 1e-8 bits)."""
        s = self.stats[:, 1]
        return _log2_rows(self.stats[:, 1:2])[:, 0] - self.stats[:, 2] / s

    def power_per_band_bits(self):
        """log2(sum_t P + eps) - max   (tfr_info.py:93)."""
        b = _log2_rows(self.power_band, float(scales.EPSILON64))
        return b - b.max(dim=1, keepdim=True).values

    def power_per_time_bits(self):
        """log2(sum_j P + eps) - max   (tfr_info.py:91)."""
        b = _log2_rows(self.power_time.to(torch.float64), float(scales.EPSILON64))
        return b - b.max(dim=1, keepdim=True).values


def styx_bank_tables(order, n, fs, dictionary_type="norm"):
    """Host float64 tables of the styx_cwt Gabor bank (styx_cwt.py:29-40,68-144):
    returns (f_hz, p_re, p_im, omega, amp, scale)."""
    f_hz = scales.log_frequency_hz_from_fft_points(fs, n, order)
    scale, omega = scales.scale_from_frequency_hz(order, f_hz, fs)
    amp_norm = (np.pi * scale ** 2) ** (-1 / 4)
    if dictionary_type == "spect":
        amp = (4 * np.pi * scale ** 2) ** (-1 / 4) * amp_norm
    elif dictionary_type == "unit":
        amp = np.ones(scale.shape)
    else:
        amp = amp_norm
    return f_hz, 0.5 / scale ** 2, np.zeros_like(scale), omega, amp, scale


class TfrPlan:
    """GPU plan for records of n samples."""

    def __init__(self, n, dtype=torch.float32, device=None, workspace_bytes=None, engine=_lib.QI_ENGINE_AUTO):
        self._lib = _lib.require_gpu()
        self.n = int(n)
        self.rdtype = _real_dtype(dtype)
        self.device = torch.device(device) if device is not None else default_device()
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.workspace_bytes = int(workspace_bytes) if workspace_bytes else 0
        self._handle = C.c_void_p()
        if os.environ.get("QI_FORCE_HIPFFT"):  # testing aid: run everything on the hipFFT engine
            engine = _lib.QI_ENGINE_HIPFFT
        desc = _lib.PlanDesc(
            n=self.n,
            dtype=_lib.QI_F64 if self.rdtype == torch.float64 else _lib.QI_F32,
            device=self.device.index,
            engine=engine,
            flags=0,
            workspace_bytes=self.workspace_bytes,
        )
        _lib.check(self._lib.qi_plan_create(C.byref(self._handle), C.byref(desc)))
        self.freq = {}  # bank name -> host band centre frequencies

    # -- sizing -------------------------------------------------------------------------------
    @staticmethod
    def workspace_for(n, n_bands, dtype, channels=1, cap_bytes=8 << 30):
        """Scratch that lets `channels` records of an n_bands panel go through in one tile."""
        esz = 16 if _real_dtype(dtype) == torch.float64 else 8
        length = 2 * n if (n & (n - 1)) == 0 else 1 << (2 * n - 2).bit_length()
        nblk = -(-n // _EPI_SPAN)
        per_chan = (n_bands + 1) * length * esz + n_bands * nblk * 32 + 4096
        build = min(n_bands, 16) * length * 16
        need = max(per_chan * channels, build, 1 << 24)
        return int(min(max(need, per_chan), max(cap_bytes, per_chan)))

    # -- tables -------------------------------------------------------------------------------
    def _stream(self):
        return _lib.stream_ptr(self.device)

    def set_gabor_bank(self, which, f_hz, p_re, p_im, omega, amp):
        keep = [_lib.darr(a) for a in (p_re, p_im, omega, amp)]
        with torch.cuda.device(self.device):
            _lib.check(
                self._lib.qi_plan_set_gabor_bank(
                    self._handle, which, len(keep[0][0]), keep[0][1], keep[1][1], keep[2][1], keep[3][1], self._stream()
                )
            )
        self.freq[which] = np.asarray(f_hz)

    def set_styx_bank(self, order, fs, dictionary_type="norm"):
        f_hz, p_re, p_im, omega, amp, _ = styx_bank_tables(order, self.n, fs, dictionary_type)
        self.set_gabor_bank(_lib.QI_BANK_STYX, f_hz, p_re, p_im, omega, amp)
        return f_hz

    def set_stx_bands(self, order, fs):
        """Band table of styx_stx.stx_complex_any_scale_pow2 (styx_stx.py:207-219,233)."""
        f_hz = scales.log_frequency_hz_from_fft_points(fs, self.n, order)
        idx = scales.stx_shift_indices(f_hz, self.n, fs)
        sigma = scales.cycles_from_order(order) / (2 * np.pi * f_hz / fs)
        ia, ip = _lib.iarr(idx)
        sa, sp = _lib.darr(sigma)
        _lib.check(self._lib.qi_plan_set_stx_bands(self._handle, len(ia), ip, sp))
        self.freq[_lib.QI_TABLE_STX] = f_hz
        self.stx_index = idx
        return f_hz

    # -- transforms ---------------------------------------------------------------------------
    # -- measurement --------------------------------------------------------------------------
    def profile(self, enable=True, stages=None, period=1):
        """Time stage launches with HIP events on the current stream (qi_plan_profile): every stage, or only the
        named ones (`stages`, names from `_lib.STAGES`), on every transform call or every `period`-th one -- each
        recorded event is a small bubble in the stream."""
        code = 1 if enable else 0
        if enable and stages is not None:
            code = 0
            for name in stages:
                code |= 1 << (_lib.STAGES.index(name) + 1)
        if enable and period > 1:
            code |= int(period) << 16
        _lib.check(self._lib.qi_plan_profile(self._handle, code))

    def stage_bands(self, stage):
        """Bands (over the styx, atoms and Stockwell tables) whose coefficients the kernels of `stage` produce."""
        k = _lib.STAGES.index(stage)
        return [int(self._lib.qi_plan_stage_bands(self._handle, which, k)) for which in (0, 1, 2)]

    def profile_read(self):
        """{stage name: (total ms, launches)} since the last read (qi_plan_profile_read)."""
        ms = (C.c_double * len(_lib.STAGES))()
        cnt = (C.c_int64 * len(_lib.STAGES))()
        _lib.check(self._lib.qi_plan_profile_read(self._handle, ms, cnt, len(_lib.STAGES)))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.STAGES)}

    def _signal(self, sig):
        if sig.dtype != self.rdtype or not sig.is_cuda or sig.device != self.device:
            sig = sig.to(device=self.device, dtype=self.rdtype)
        sig = sig.contiguous()
        if sig.dim() != 2 or sig.shape[1] != self.n:
            raise ValueError(f"signal must be [channels, {self.n}], got {tuple(sig.shape)}")
        return sig

    def _outputs(self, which, n_ch, coef, bits, reductions, power_scale, eps, out, reduced_out):
        """The TfrResult of one transform (new buffers, or those of `out`) and its C-ABI descriptor."""
        f_hz = self.freq.get(which)
        if f_hz is None:
            raise _lib.QiError("band table not set on this plan")
        n_b = len(f_hz)
        dev = self.device
        if out is not None:  # reuse the buffers of an earlier call of the same shape
            res = out
            if (res.coef is not None and res.coef.shape[0] != n_ch) or (res.stats is not None and res.stats.shape[0] != n_ch):
                raise ValueError("out= buffers were made for another channel count")
        else:
            res = TfrResult(frequency_hz=f_hz, power_scale=power_scale)
            if coef:
                res.coef = torch.empty((n_ch, n_b, self.n), dtype=_complex_of(self.rdtype), device=dev)
            if bits:
                res.bits = torch.empty((n_ch, n_b, self.n), dtype=self.rdtype, device=dev)
            if reductions:
                # one buffer for the whole reduced product (the message of dist.gather_reduced); the three outputs
                # are views into it
                from .dist import reduced_slots

                slots = reduced_slots(n_ch, n_b, self.n, self.rdtype)
                if reductions == "band":
                    # band powers and statistics only, no per-time marginal: the kernels then write (and the tail reads) no
                    # per-time planes at all -- what a streaming job that keeps no per-time power asks for (stream.py)
                    if reduced_out is not None:
                        raise ValueError('reductions="band" has no gather layout: reduced_out does not apply')
                    small = torch.empty(n_ch * (n_b + 4), dtype=torch.float64, device=dev)
                    res.power_band = small[: n_ch * n_b].view(n_ch, n_b)
                    res.stats = small[n_ch * n_b :].view(n_ch, 4)
                    slots = None
                elif reduced_out is not None:  # caller-provided slice (e.g. of one buffer for several transforms)
                    if reduced_out.dtype != torch.float64 or reduced_out.numel() != slots or not reduced_out.is_contiguous():
                        raise ValueError(f"reduced_out must be a contiguous float64 tensor of {slots} elements")
                    res.reduced = reduced_out
                else:
                    res.reduced = torch.empty(slots, dtype=torch.float64, device=dev)
                if slots is not None:
                    o1 = res.reduced.numel() - n_ch * (n_b + 4)
                    o2 = o1 + n_ch * n_b
                    res.power_time = res.reduced[:o1].view(self.rdtype)[: n_ch * self.n].view(n_ch, self.n)
                    res.power_band = res.reduced[o1:o2].view(n_ch, n_b)
                    res.stats = res.reduced[o2:].view(n_ch, 4)
        desc = _lib.TfrOut(
            coef=_lib.ptr(res.coef),
            bits=_lib.ptr(res.bits),
            power_band=_lib.ptr(res.power_band),
            power_time=_lib.ptr(res.power_time),
            stats=_lib.ptr(res.stats),
            power_scale=float(power_scale),
            eps=float(eps),
        )
        return res, desc

    def _run(self, which, sig, coef, bits, reductions, power_scale, eps, out=None, reduced_out=None):
        sig = self._signal(sig)
        n_ch = sig.shape[0]
        res, desc = self._outputs(which, n_ch, coef, bits, reductions, power_scale, eps, out, reduced_out)
        with torch.cuda.device(self.device):
            if which == _lib.QI_TABLE_STX:
                rc = self._lib.qi_stx(self._handle, _lib.ptr(sig), n_ch, C.byref(desc), self._stream())
            else:
                rc = self._lib.qi_cwt(self._handle, which, _lib.ptr(sig), n_ch, C.byref(desc), self._stream())
        _lib.check(rc)
        return res

    def cwt_stx(self, sig, coef=True, bits=False, reductions=False, power_scale=1.0, eps=0.0, out=None, reduced_out=None):
        """The styx CWT and the Stockwell transform of the same records in one call (qi_cwt_stx): (cwt, stx) results,
        equal to `cwt(...)` then `stx(...)` to within float rounding; `out` / `reduced_out` are pairs."""
        sig = self._signal(sig)
        n_ch = sig.shape[0]
        out = out or (None, None)
        reduced_out = reduced_out or (None, None)
        res_c, desc_c = self._outputs(_lib.QI_BANK_STYX, n_ch, coef, bits, reductions, power_scale, eps, out[0], reduced_out[0])
        res_s, desc_s = self._outputs(_lib.QI_TABLE_STX, n_ch, coef, bits, reductions, power_scale, eps, out[1], reduced_out[1])
        with torch.cuda.device(self.device):
            rc = self._lib.qi_cwt_stx(self._handle, _lib.QI_BANK_STYX, _lib.ptr(sig), n_ch, C.byref(desc_c),
                                      C.byref(desc_s), self._stream())
        _lib.check(rc)
        return res_c, res_s

    def cwt(self, sig, coef=True, bits=False, reductions=False, power_scale=1.0, eps=0.0, out=None, reduced_out=None):
        return self._run(_lib.QI_BANK_STYX, sig, coef, bits, reductions, power_scale, eps, out, reduced_out)

    def cwt_atoms(self, sig, coef=True, bits=False, reductions=False, power_scale=1.0, eps=0.0, out=None, reduced_out=None):
        return self._run(_lib.QI_BANK_ATOMS, sig, coef, bits, reductions, power_scale, eps, out, reduced_out)

    def stx(self, sig, coef=True, bits=False, reductions=False, power_scale=1.0, eps=0.0, out=None, reduced_out=None):
        return self._run(_lib.QI_TABLE_STX, sig, coef, bits, reductions, power_scale, eps, out, reduced_out)

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qi_plan_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlanRing:
    """Independent records, one call each: `depth` plans with the same tables on `depth` streams, used in turn.

    A call of one or two records spends a fifth of its time in short launches (forward transform, coarse stage, tail) that
    leave most of the chip idle; inside ONE call they cannot be hidden (cross-stream events cost more than the overlap
    saves), between INDEPENDENT calls they can: the short launches of one call run under the long launches of the other
    (configs[1]: 434 000 against 387 000 Mpoints/s).  No cross-stream synchronisation happens here; every result carries
    the event that follows its launches.

        ring = PlanRing(n, torch.float32, setup=lambda p: (p.set_styx_bank(3, fs), p.set_stx_bands(3, fs)))
        for record in records:                       # [1, n] tensors on the device
            res_c, res_s, done = ring.cwt_stx(record, coef=True, reductions=True)
            ...                                      # done.synchronize() / stream.wait_event(done) before reading

    Results of a slot are overwritten `depth` calls later (they are the slot's `out=` buffers): consume or copy them
    before then.  `record` must not be modified before `done` either; it may be dropped (its memory is tied to the
    slot's stream with `record_stream`)."""

    def __init__(self, n, dtype=torch.float32, device=None, workspace_bytes=None, setup=None, depth=2,
                 engine=_lib.QI_ENGINE_AUTO, wait_input=True):
        self.wait_input = wait_input  # False: the caller guarantees the records are ready (no event on its stream)
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.plans = [TfrPlan(n, dtype, device, workspace_bytes, engine) for _ in range(depth)]
        for pl in self.plans:
            if setup is not None:
                setup(pl)
        self.streams = [torch.cuda.Stream(device=self.plans[0].device) for _ in range(depth)]
        self._outs = [dict() for _ in range(depth)]
        self._turn = 0

    def _call(self, name, sig, **kw):
        j = self._turn
        self._turn = (j + 1) % len(self.plans)
        stream = self.streams[j]
        if self.wait_input:  # the record was produced on the caller's stream
            stream.wait_stream(torch.cuda.current_stream(self.plans[j].device))
        key = (name, tuple(sig.shape), tuple(sorted(kw.items())))
        if sig.is_cuda:
            sig.record_stream(stream)  # the caller may drop the record: its memory is not reused before this stream is done
        with torch.cuda.stream(stream):
            res = getattr(self.plans[j], name)(sig, out=self._outs[j].get(key), **kw)
            self._outs[j][key] = res
            done = torch.cuda.Event()
            done.record(stream)
        return res, done

    def cwt_stx(self, sig, **kw):
        (res_c, res_s), done = self._call("cwt_stx", sig, **kw)
        return res_c, res_s, done

    def cwt(self, sig, **kw):
        return self._call("cwt", sig, **kw)

    def stx(self, sig, **kw):
        return self._call("stx", sig, **kw)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.synchronize()
        for pl in self.plans:
            pl.close()
        self.plans = []


# ---- small LRU of plans for the reference-signature wrappers (each call there is one record) ----
_PLANS = collections.OrderedDict()
_MAX_PLANS = 3


def cached_plan(key, factory):
    plan = _PLANS.pop(key, None)
    if plan is None:
        plan = factory()
        while len(_PLANS) >= _MAX_PLANS:
            _, old = _PLANS.popitem(last=False)
            old.close()
    _PLANS[key] = plan
    return plan


def clear_plans():
    while _PLANS:
        _, old = _PLANS.popitem()
        old.close()


def gabor_atoms(n, p_re, p_im, omega, amp, device=None, x=None):
    """[B, n] complex128 device tensor of time-domain atoms (qi_gabor_atoms); x: the sample positions [n] (float64,
    in samples relative to the atom's centre) when they are not the centred uniform ones (qi_gabor_atoms_at)."""
    lib = _lib.require_gpu()
    dev = torch.device(device) if device is not None else default_device()
    keep = [_lib.darr(a) for a in (p_re, p_im, omega, amp)]
    out = torch.empty((len(keep[0][0]), n), dtype=torch.complex128, device=dev)
    if x is not None:
        if np.shape(x) != (n,):
            raise ValueError(f"x must hold the {n} sample positions of the atoms, got shape {np.shape(x)}")
        xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
        with torch.cuda.device(dev):
            _lib.check(
                lib.qi_gabor_atoms_at(dev.index if dev.index is not None else torch.cuda.current_device(), n,
                                      len(keep[0][0]), keep[0][1], keep[1][1], keep[2][1], keep[3][1], _lib.ptr(xd),
                                      _lib.ptr(out), _lib.stream_ptr(dev))
            )
        return out
    with torch.cuda.device(dev):
        _lib.check(
            lib.qi_gabor_atoms(
                dev.index if dev.index is not None else torch.cuda.current_device(),
                n,
                len(keep[0][0]),
                keep[0][1],
                keep[1][1],
                keep[2][1],
                keep[3][1],
                _lib.ptr(out),
                _lib.stream_ptr(dev),
            )
        )
    return out


# What the reference-signature wrappers hand back to NumPy callers for float32 records.  The reference returns
# complex128 panels (float64 bits) whatever the record's dtype (styx_cwt.py:195-198, styx_stx.py:228,
# cwt_atoms.py:408): "reference" computes in float32 and widens on the way out, so a drop-in caller sees the
# reference's dtypes; "native" keeps complex64 / float32 (half the host memory and copy time).  CUDA tensors in ->
# tensors out are never widened.
NUMPY_RESULT_DTYPE = "reference"


def finish(result_tensor, was_numpy, was_1d, widen=False):
    """Undo as_signal's batching / device move on an output.  widen: a panel the reference would return in double
    precision (see NUMPY_RESULT_DTYPE).  NumPy callers get an array backed by page-locked memory (the copy from the
    device then runs at PCIe speed instead of through a pageable staging buffer); the widening runs on the device."""
    if result_tensor is None:
        return None
    t = result_tensor[0] if was_1d else result_tensor
    if not was_numpy:
        return t
    t = t.contiguous()
    if widen and NUMPY_RESULT_DTYPE == "reference" and t.dtype in (torch.complex64, torch.float32) and t.numel() > 0:
        wide = torch.empty(t.shape, dtype=torch.complex128 if t.dtype == torch.complex64 else torch.float64, device=t.device)
        src = torch.view_as_real(t) if t.is_complex() else t
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().qi_widen(t.device.index, _lib.ptr(src), _lib.ptr(wide), src.numel(), _lib.stream_ptr(t.device)))
        t = wide
    nbytes = t.numel() * t.element_size()
    if nbytes >= PINNED_RESULT_MAX_BYTES:
        return _staged_copy(t)  # bounded page-locked memory: two 64 MiB staging buffers, pageable result
    if nbytes >= (1 << 20):
        # the result itself in page-locked memory: one copy at PCIe speed.  PyTorch's pinned allocator keeps freed blocks
        # page-locked, so what it has cached is handed back to the system once it exceeds the cap.
        _trim_pinned_cache()
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t)
        return host.numpy()
    return t.cpu().numpy()


PINNED_RESULT_MAX_BYTES = 4 << 30  # larger results are staged through two fixed buffers into pageable memory
STAGE_PIECE_BYTES = 64 << 20  # bytes per staged piece (<= the 64 MiB staging buffers; tests lower it to exercise many pieces)


def _trim_pinned_cache():
    """Return cached (freed, still page-locked) host blocks to the system when they exceed PINNED_RESULT_MAX_BYTES."""
    try:
        stats = torch.cuda.memory.host_memory_stats()
        cached = int(stats.get("allocated_bytes.current", 0)) - int(stats.get("active_bytes.current", 0))
        if cached > PINNED_RESULT_MAX_BYTES:
            torch._C._host_emptyCache()
    except Exception:  # (older PyTorch: no host allocator statistics -- nothing to trim with)
        pass


_STAGE_BYTES = 64 << 20
_STAGE = []  # two page-locked staging buffers, made once (PyTorch's pinned allocator never unpins what it freed)
_STAGE_LOCK = threading.Lock()  # the two buffers are shared by every caller of finish(): one staged copy at a time
_NUMPY_OF = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64, torch.complex128: np.complex128,
             torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}


def _staged_copy(t):
    """Device tensor -> pageable NumPy array through two fixed page-locked buffers: the device-to-host copy of piece k + 1
    runs while piece k is copied out of its staging buffer, and no page-locked memory grows with the result (an order-12
    panel is 1.6 GB and more once widened)."""
    if t.dtype not in _NUMPY_OF:
        raise TypeError(f"no NumPy dtype for a {t.dtype} result")
    flat = t.reshape(-1).view(torch.uint8) if not t.is_complex() else torch.view_as_real(t).reshape(-1).view(torch.uint8)
    out = np.empty(t.shape, dtype=_NUMPY_OF[t.dtype])
    dst = torch.from_numpy(out.reshape(-1).view(np.uint8))
    total = flat.numel()
    stream = torch.cuda.current_stream(t.device)
    events = [torch.cuda.Event(), torch.cuda.Event()]
    piece = min(_STAGE_BYTES, max(int(STAGE_PIECE_BYTES), 4096))
    pieces = [(o, min(piece, total - o)) for o in range(0, total, piece)]
    with _STAGE_LOCK:
        if not _STAGE:
            _STAGE.extend(torch.empty(_STAGE_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2))
        for k, (o, m) in enumerate(pieces):
            _STAGE[k & 1][:m].copy_(flat[o : o + m], non_blocking=True)
            events[k & 1].record(stream)
            if k:
                po, pm = pieces[k - 1]
                events[(k - 1) & 1].synchronize()
                dst[po : po + pm].copy_(_STAGE[(k - 1) & 1][:pm])
        po, pm = pieces[-1]
        events[(len(pieces) - 1) & 1].synchronize()
        dst[po : po + pm].copy_(_STAGE[(len(pieces) - 1) & 1][:pm])
    return out
