"""ctypes binding of libqi_tfr.so (include/qi_tfr.h).  There is no CPU fallback: if the
library is missing or no HIP device is present, every transform raises."""
import ctypes as C
import os

import torch  # first: its bundled HIP runtime must be the one libqi_tfr.so resolves against

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QI_TFR_LIB") or os.path.join(_HERE, "libqi_tfr.so")  # QI_TFR_LIB: diagnostic builds

QI_F32, QI_F64 = 0, 1
QI_BANK_STYX, QI_BANK_ATOMS, QI_TABLE_STX = 0, 1, 2
QI_ENGINE_AUTO, QI_ENGINE_HIPFFT, QI_ENGINE_NATIVE = 0, 1, 2
STAGES = ("forward", "multiply", "inverse", "epilogue", "pass1", "pass2", "block", "zoom", "zoom_coarse")


class QiError(RuntimeError):
    pass


class PlanDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("dtype", C.c_int32),
        ("device", C.c_int32),
        ("engine", C.c_int32),
        ("flags", C.c_int32),
        ("workspace_bytes", C.c_int64),
    ]


class TfrOut(C.Structure):
    _fields_ = [
        ("coef", C.c_void_p),
        ("bits", C.c_void_p),
        ("power_band", C.c_void_p),
        ("power_time", C.c_void_p),
        ("stats", C.c_void_p),
        ("power_scale", C.c_double),
        ("eps", C.c_double),
    ]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_i64, _i32, _int, _dbl = C.c_int64, C.c_int32, C.c_int, C.c_double

# name -> (restype, argtypes); mirrors include/qi_tfr.h one to one
PROTOTYPES = {
    "qi_abi_version": (_int, []),
    "qi_last_error": (C.c_char_p, []),
    "qi_device_info": (_int, [_int, C.c_char_p, C.c_size_t, _I64, C.POINTER(_i32)]),
    "qi_plan_create": (_int, [C.POINTER(_P), C.POINTER(PlanDesc)]),
    "qi_plan_destroy": (_int, [_P]),
    "qi_plan_set_gabor_bank": (_int, [_P, _int, _i32, _D, _D, _D, _D, _P]),
    "qi_gabor_atoms": (_int, [_int, _i64, _i32, _D, _D, _D, _D, _P, _P]),
    "qi_gabor_atoms_at": (_int, [_int, _i64, _i32, _D, _D, _D, _D, _P, _P, _P]),
    "qi_plan_set_stx_bands": (_int, [_P, _i32, _I64, _D]),
    "qi_plan_bands": (_i64, [_P, _int]),
    "qi_plan_stage_bands": (_i64, [_P, _int, _int]),
    "qi_plan_profile": (_int, [_P, _int]),
    "qi_plan_profile_read": (_int, [_P, _D, _I64, _i32]),
    "qi_cwt": (_int, [_P, _int, _P, _i64, C.POINTER(TfrOut), _P]),
    "qi_stx": (_int, [_P, _P, _i64, C.POINTER(TfrOut), _P]),
    "qi_cwt_stx": (_int, [_P, _int, _P, _i64, C.POINTER(TfrOut), C.POINTER(TfrOut), _P]),
    "qi_stft_segments": (_i64, [_i64, _i64, _i64]),
    "qi_stft_scratch_bytes": (_i64, [_int, _i64, _i64, _i64, _i64, _i64]),
    "qi_stft": (_int, [_int, _int, _P, _i64, _i64, _P, _i64, _i64, _i64, _dbl, _P, _P, _dbl, _P, _i64, _P]),
    "qi_welch_scratch_bytes": (_i64, [_int, _i64, _i64, _i64, _i64, _i64]),
    "qi_welch": (_int, [_int, _int, _P, _i64, _i64, _P, _i64, _i64, _i64, _dbl, _P, _P, _i64, _P]),
    "qi_power_marginals": (_int, [_int, _int, _P, _i64, _i64, _i64, _P, _P, _P, _P, _i64, _P]),
    "qi_power_marginals_scratch_bytes": (_i64, [_i64, _i64, _i64]),
    "qi_log2_offset": (_int, [_int, _int, _P, _P, _i64, _i64, _dbl, _P, _P]),
    "qi_widen": (_int, [_int, _P, _P, _i64, _P]),
    "qi_log2_abs": (_int, [_int, _int, _P, _int, _P, _i64, _dbl, _P]),
    "qi_shannon_panel": (_int, [_int, _int, _P, _P, _int, _i64, _i64, _i64, _dbl, _P, _P, _P, _P, _P]),
    "qi_sliding_scratch_bytes": (_i64, [_int, _i64, _i64, _i64]),
    "qi_sliding_stft": (_int, [_int, _int, _P, _i64, _i64, _P, _i64, _i64, _i64, _i64, _i64, _int, _int, _i64, _P, _P, _int, _P, _i64, _P]),
    "qi_sliding_istft": (_int, [_int, _int, _P, _i64, _P, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _P, _P, _i64, _P]),
    "qi_shannon_1d": (_int, [_int, _int, _P, _i64, _i64, _P, _P, _P, _P, _P]),
    "qi_shannon_scratch_bytes": (_i64, [_int, _i64, _i64]),
    "qi_shannon_tdr": (_int, [_int, _int, _P, _i64, _i64, _P, _P, _P, _i64, _P]),
    "qi_shannon_fft": (_int, [_int, _int, _P, _i64, _i64, _P, _P, _P, _P, _i64, _P]),
}

_lib = None


def _hip_runtimes_mapped():
    try:
        with open("/proc/self/maps") as fh:
            return sorted({ln.split()[-1] for ln in fh if "libamdhip64" in ln})
    except OSError:
        return []


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QiError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise QiError(f"two HIP runtimes are mapped ({rts}); libqi_tfr.so must share torch's")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().qi_last_error()
        raise QiError(f"libqi_tfr error {rc}: {msg.decode() if msg else '?'}")


def require_gpu():
    if not torch.cuda.is_available():
        raise QiError("no HIP device visible: the TFR transforms run only on the GPU (no CPU fallback)")
    return load()


def darr(a):
    import numpy as np

    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_D)


def iarr(a):
    import numpy as np

    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_I64)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
