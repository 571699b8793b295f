"""
quantum-inferno TFR hot path, MI355X-native.

Drop-in for the FFT-based time-frequency stack of ISLA-UH/quantum-inferno: same module and
function names (styx_fft, styx_cwt, styx_stx, cwt_atoms, tfr_info over scales_dyadic bands),
computed by hand-written HIP kernels for gfx950 behind a C ABI (include/qi_tfr.h).
There is no CPU fallback: transforms raise if libqi_tfr.so or a HIP device is missing.
"""
from . import scales_dyadic  # noqa: F401  (host tables; importable without a GPU)
from . import utilities  # noqa: F401
from . import _lib, engine  # noqa: F401
from . import styx_fft, styx_cwt, styx_stx, cwt_atoms, tfr_info  # noqa: F401
from .engine import PlanRing, TfrPlan, TfrResult  # noqa: F401

__version__ = "0.1.0"
