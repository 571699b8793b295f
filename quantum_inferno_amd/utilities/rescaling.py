"""log2 rescaling (mirror of quantum_inferno/utilities/rescaling.py:13-28).  Panels produced by
the transforms get their bits from the fused GPU epilogue (qi_tfr_out.bits); this module is the
host scalar / small-array helper the reference also exposes."""
from typing import Union

import numpy as np

from ..scales_dyadic import get_epsilon


def to_log2_with_epsilon(x: Union[np.ndarray, float, list]) -> Union[np.ndarray, float]:
    """log2(|x| + eps): amplitude bits, complex input allowed (ref rescaling.py:13-20).  NumPy / scalar / host-tensor input: a float64
    NumPy result as in the reference.  A CUDA tensor goes through the library's kernel (qi_log2_abs) and keeps its precision: float32 /
    complex64 in -> float32 bits (the reference, NumPy only, would return float64; cast the input to double to get that)."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(x, torch.Tensor) and x.is_cuda:
        from .. import tfr_info  # device tensors: the library's log2 kernel (qi_log2_abs), not a PyTorch expression

        return tfr_info.log2_abs(x, float(get_epsilon()))
    if torch is not None and isinstance(x, torch.Tensor):  # a host tensor is NumPy data: float64 array out, as the signature says
        x = x.detach().numpy()
    return np.log2(np.abs(x) + get_epsilon())


def is_power_of_two(n: int) -> bool:
    """True for positive powers of two (ref rescaling.py:23-28)."""
    return n > 0 and not (n & (n - 1))
