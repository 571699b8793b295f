"""log2 rescaling (mirror of quantum_inferno/utilities/rescaling.py:13-28).  Panels produced by
the transforms get their bits from the fused GPU epilogue (qi_tfr_out.bits); this module is the
host scalar / small-array helper the reference also exposes."""
from typing import Union

import numpy as np

from ..scales_dyadic import get_epsilon


def to_log2_with_epsilon(x: Union[np.ndarray, float, list]) -> Union[np.ndarray, float]:
    """log2(|x| + eps): amplitude bits, complex input allowed (ref rescaling.py:13-20)."""
    try:
        import torch

        if isinstance(x, torch.Tensor):
            return torch.log2(torch.abs(x) + float(get_epsilon()))
    except ImportError:  # pragma: no cover
        pass
    return np.log2(np.abs(x) + get_epsilon())


def is_power_of_two(n: int) -> bool:
    """True for positive powers of two (ref rescaling.py:23-28)."""
    return n > 0 and not (n & (n - 1))
