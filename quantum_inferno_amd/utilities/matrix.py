"""Row / column broadcast multiplies of a [frequency x time] panel (semantics of
quantum_inferno/utilities/matrix.py:89-134).  The reference materialises a tiled copy with
np.tile; here it is a broadcast, and inside the GPU kernels (qi_shannon_panel) an in-register one."""
import numpy as np


def d0tile_x_d0d1(d0, d0d1):
    """Multiply each row j of d0d1 by d0[j] (ref matrix.py:89-110)."""
    d0d1 = np.asarray(d0d1)
    d0 = np.asarray(d0)
    if d0d1.ndim == 1:
        if d0.shape not in ((), d0d1.shape):
            raise TypeError(f"Cannot handle an array of shape {d0.shape}.")
        return d0 * d0d1
    if d0d1.ndim != 2 or d0.shape != (d0d1.shape[0],):
        raise TypeError(f"Cannot handle an array of shape {d0.shape}.")
    return d0[:, None] * d0d1


def d1tile_x_d0d1(d1, d0d1):
    """Multiply each column t of d0d1 by d1[t] (ref matrix.py:113-134)."""
    d0d1 = np.asarray(d0d1)
    d1 = np.asarray(d1)
    if d0d1.ndim == 1:
        if d1.shape not in ((), d0d1.shape):
            raise TypeError(f"Cannot handle an array of shape {d1.shape}.")
        return d1 * d0d1
    if d0d1.ndim != 2 or d1.shape != (d0d1.shape[1],):
        raise TypeError(f"Cannot handle an array of shape {d1.shape}.")
    return d1[None, :] * d0d1
