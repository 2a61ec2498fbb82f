import numpy as _np


def convert_dtype(dtype, namespace):
    """NumPy dtype of `dtype` (the real helper maps between backends' dtype objects)."""
    return _np.dtype(dtype).type
